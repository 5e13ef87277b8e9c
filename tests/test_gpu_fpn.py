"""The callers either side of the encoder on the GPU, through the C ABI:
``DiTWithFPN`` (lateral 1x1 on tokens, NHWC top-down merge, implicit-GEMM 3x3, LastLevelMaxPool view) against the torch
restatement of torchvision's FeaturePyramidNetwork in the REFERENCE's order of operations (oracle/fpn_oracle_torch.py -
parity unpinned with respect to torchvision itself, its source is not available offline); the detector input transform
for ragged fp32 / fp16 image lists against ``F.interpolate``; the fp16 entry of the encoder."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")
import torch.nn.functional as F                      # noqa: E402

from layoutdit_amd import config as cfgs, ops, synth     # noqa: E402
from layoutdit_amd.modeling import DetectorInputTransform, DiTEncoder, DiTWithFPN    # noqa: E402
from oracle import oracle                            # noqa: E402
from oracle.fpn_oracle_torch import backbone_maps, backbone_maps_t, fpn_forward, fpn_forward_t     # noqa: E402
from tests.util import rel_l2                        # noqa: E402

DEV = "cuda:0"


def _rand(seed, *shape, scale=1.0):
    n = int(np.prod(shape))
    return (scale * synth.normal(seed, 9, n)).astype(np.float32).reshape(shape)


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(2, 7, 7, 256, 256), (1, 56, 56, 256, 256), (3, 5, 9, 64, 96), (1, 1, 1, 32, 32),
                                            (2, 28, 28, 256, 256)])
def test_conv3x3_nhwc_implicit_gemm(B, H, W, Cin, Cout):
    x, w, b = _rand(1, B, H, W, Cin), _rand(2, Cout, Cin, 3, 3, scale=0.05), _rand(3, Cout, scale=0.1)
    y = ops.conv3x3_nhwc(torch.from_numpy(x).to(DEV), torch.from_numpy(w).permute(0, 2, 3, 1).contiguous().to(DEV),
                         torch.from_numpy(b).to(DEV))
    ref = F.conv2d(torch.from_numpy(x).double().permute(0, 3, 1, 2), torch.from_numpy(w).double(), torch.from_numpy(b).double(),
                   padding=1).permute(0, 2, 3, 1)
    assert tuple(y.shape) == (B, H, W, Cout)
    assert rel_l2(y.cpu().numpy(), ref.numpy()) < 1e-5


@pytest.mark.parametrize("gh,gw,scale,with_top", [(14, 14, 0.5, False), (14, 14, 1.0, True), (14, 14, 2.0, True),
                                                  (14, 14, 4.0, True), (6, 4, 2.0, True), (5, 7, 0.5, False)])
def test_fpn_merge_level(gh, gw, scale, with_top):
    B, Ch = 2, 256
    lat = _rand(4, B, gh * gw + 1, Ch)
    oh, ow = int(gh * scale), int(gw * scale)
    th, tw = max(oh // 2, 1), max(ow // 2, 1)
    top = _rand(5, B, th, tw, Ch) if with_top else None
    got = ops.fpn_merge(torch.from_numpy(lat).to(DEV), gh, gw, scale, None if top is None else torch.from_numpy(top).to(DEV))
    m = torch.from_numpy(lat).double()[:, 1:, :].permute(0, 2, 1).reshape(B, Ch, gh, gw)
    if scale != 1.0:
        m = F.interpolate(m, scale_factor=scale, mode="bilinear", align_corners=False)
    if top is not None:
        m = m + F.interpolate(torch.from_numpy(top).double().permute(0, 3, 1, 2), size=(oh, ow), mode="nearest")
    assert tuple(got.shape) == (B, oh, ow, Ch)
    assert rel_l2(got.cpu().numpy(), m.permute(0, 2, 3, 1).numpy()) < 1e-6


def _fpn_weights(m):
    return {k[len("fpn."):]: v.detach().cpu().numpy() for k, v in m.state_dict().items() if k.startswith("fpn.")}


@pytest.mark.parametrize("geom,size,B", [("tiny", 224, 2), ("micro", 64, 3)])
def test_dit_with_fpn_vs_reference_order_oracle(geom, size, B):
    cfg = cfgs.GEOMETRIES[geom]()
    w = synth.synth_weights(cfg, 1)
    x = synth.synth_images(B, size, size, seed=1234)
    torch.manual_seed(0)
    m = DiTWithFPN(config=cfg)
    m.backbone.dit.load_numpy(w)
    with torch.no_grad():
        for p in m.fpn.parameters():                       # non-trivial biases (torchvision's init zeroes them)
            if p.dim() == 1:
                p.normal_(0.0, 0.05)
    m = m.to(DEV).eval()
    with torch.no_grad():
        feats = m(torch.from_numpy(x).to(DEV))
    g = size // 16
    assert list(feats) == ["p2", "p3", "p4", "p5", "pool"] and m.out_channels == 256
    taps, _ = oracle.vit_forward(cfg, w, x)
    ref = fpn_forward(backbone_maps(taps, g, g), _fpn_weights(m))
    for k, r in ref.items():
        a = feats[k]
        assert tuple(a.shape) == r.shape, k
        assert rel_l2(a.cpu().numpy(), r) < 2e-5, k
    assert tuple(feats["pool"].shape[-2:]) == ((g // 2 + 1) // 2, (g // 2 + 1) // 2)
    # checkpoint surface: torchvision's FPN key names
    keys = set(m.state_dict())
    assert {"fpn.inner_blocks.0.0.weight", "fpn.layer_blocks.3.0.bias"} <= keys
    assert tuple(m.state_dict()["fpn.layer_blocks.0.0.weight"].shape) == (256, 256, 3, 3)


@pytest.mark.parametrize("gh,gw,scale,with_top", [(14, 14, 0.5, False), (14, 14, 1.0, True), (14, 14, 2.0, True), (14, 14, 4.0, True),
                                                  (6, 4, 2.0, True), (5, 7, 0.5, False), (5, 7, 1.0, True), (3, 5, 4.0, True)])
def test_fpn_merge_adjoint(gh, gw, scale, with_top):
    """ldit_fpn_merge_bwd_f32 against torch autograd of the forward's own definition (bilinear rescale of the lateral tokens
    + nearest top-down add), odd grids included (a 5 x 7 top feeds a 2 x 3 ... the clamp of the nearest index)."""
    B, Ch = 2, 64
    oh, ow = int(gh * scale), int(gw * scale)
    th, tw = max(oh // 2, 1), max(ow // 2, 1)
    lat = torch.from_numpy(_rand(4, B, gh * gw + 1, Ch)).double().requires_grad_(True)
    top = torch.from_numpy(_rand(5, B, th, tw, Ch)).double().requires_grad_(True) if with_top else None
    g = _rand(6, B, oh, ow, Ch)
    m = lat[:, 1:, :].permute(0, 2, 1).reshape(B, Ch, gh, gw)
    if scale != 1.0:
        m = F.interpolate(m, scale_factor=scale, mode="bilinear", align_corners=False)
    if top is not None:
        m = m + F.interpolate(top.permute(0, 3, 1, 2), size=(oh, ow), mode="nearest")
    (m.permute(0, 2, 3, 1) * torch.from_numpy(g).double()).sum().backward()
    seed = _rand(7, B, th, tw, Ch)                                   # d_top is accumulated into: start from a non-zero buffer
    d_top = torch.from_numpy(seed.copy()).to(DEV) if with_top else None
    d_lat = ops.fpn_merge_bwd(torch.from_numpy(g).to(DEV), gh, gw, scale, d_top=d_top)
    assert rel_l2(d_lat.cpu().numpy(), lat.grad.numpy()) < 1e-6
    assert float(d_lat[:, 0].abs().max()) == 0.0
    if with_top:
        assert rel_l2(d_top.cpu().numpy() - seed, top.grad.numpy()) < 1e-5


def test_fpn_backward_building_blocks():
    """colsum, the zero-padded bf16 copy, the 3x3 dgrad (same GEMM, flipped weight) and the nine-tap bf16 wgrad on padded
    NHWC operands - each against float64 torch (the wgrad on the bf16-rounded operands it actually multiplies)."""
    B, H, W, Ch = 2, 6, 9, 64
    x, dy = _rand(50, B, H, W, Ch), _rand(51, B, H, W, Ch)
    wt = _rand(52, Ch, Ch, 3, 3, scale=0.05)
    xd, dyd = torch.from_numpy(x).to(DEV), torch.from_numpy(dy).to(DEV)
    assert rel_l2(ops.colsum(dyd.view(-1, Ch)).cpu().numpy(), dy.reshape(-1, Ch).astype(np.float64).sum(0)) < 1e-6
    big = torch.from_numpy(_rand(53, 3000, 96)).to(DEV)
    assert rel_l2(ops.colsum(big).cpu().numpy(), big.cpu().double().sum(0).numpy()) < 1e-6
    slack = W + 3
    xp = ops.pad_nhwc_bf16(xd, slack_rows=slack)
    ref_pad = F.pad(torch.from_numpy(x), (0, 0, 1, 1, 1, 1)).to(torch.bfloat16).view(-1, Ch)
    assert torch.equal(xp[slack: slack + ref_pad.shape[0]].cpu(), ref_pad)
    assert float(xp[:slack].abs().max()) == 0.0 and float(xp[slack + ref_pad.shape[0]:].abs().max()) == 0.0
    # autograd reference of y = conv3x3(x, w)
    xr = torch.from_numpy(x).double().permute(0, 3, 1, 2).requires_grad_(True)
    wr = torch.from_numpy(wt).double().requires_grad_(True)
    (F.conv2d(xr, wr, padding=1) * torch.from_numpy(dy).double().permute(0, 3, 1, 2)).sum().backward()
    from layoutdit_amd.modeling.dit_fpn import _flip_ihwo
    dx = ops.conv3x3_nhwc(dyd, _flip_ihwo(torch.from_numpy(wt).to(DEV)))
    assert rel_l2(dx.cpu().numpy(), xr.grad.permute(0, 2, 3, 1).numpy()) < 1e-5
    dyp = ops.pad_nhwc_bf16(dyd)
    taps = []
    for ky in range(3):
        for kx in range(3):
            shift = (ky - 1) * (W + 2) + (kx - 1)
            taps.append(ops.wgrad_bf16(dyp, xp, dyp.shape[0], w_row_offset=slack + shift))
    dw = torch.stack(taps, dim=2).view(Ch, Ch, 3, 3)
    xb = torch.from_numpy(x).to(torch.bfloat16).double().permute(0, 3, 1, 2).requires_grad_(False)
    wr2 = torch.from_numpy(wt).double().requires_grad_(True)
    (F.conv2d(xb, wr2, padding=1) * torch.from_numpy(dy).to(torch.bfloat16).double().permute(0, 3, 1, 2)).sum().backward()
    assert rel_l2(dw.cpu().numpy(), wr2.grad.numpy()) < 1e-5         # exact up to fp32 accumulation on the rounded operands
    assert rel_l2(dw.cpu().numpy(), wr.grad.numpy()) < 1e-2          # and within bf16 rounding of the fp32 problem


@pytest.mark.parametrize("enc_dtype", ["f32", "bf16"])
def test_dit_with_fpn_trains_like_the_reference(enc_dtype):
    """ref dit_backbone.py:87-90 under ref trainer.py:169-178: loss.backward() through self.fpn(feats) and the encoder.  Every FPN
    parameter gradient and the gradients reaching the encoder against float64 autograd of the reference-order restatement
    (oracle/fpn_oracle_torch.py on top of oracle/vit_oracle_torch.py: parity unpinned with respect to torchvision)."""
    from oracle.vit_oracle_torch import train_reference
    cfg = cfgs.vit_micro()
    cfg.drop_path_rate = 0.0
    cfg.taps = [1, 1, 2, 3]
    w = synth.synth_weights(cfg, 3)
    B, size, g = 2, 64, 4
    x = synth.synth_images(B, size, size, seed=5, kind="uniform")
    torch.manual_seed(0)
    m = DiTWithFPN(config=cfg, compute_dtype=enc_dtype)
    m.backbone.dit.load_numpy(w)
    with torch.no_grad():
        for p in m.fpn.parameters():
            if p.dim() == 1:
                p.normal_(0.0, 0.05)
    m = m.to(DEV).train()
    feats = m(torch.from_numpy(x).to(DEV))
    assert list(feats) == ["p2", "p3", "p4", "p5", "pool"]
    ws = {k: _rand(60 + i, *f.shape, scale=1.0 / np.sqrt(f.numel())) for i, (k, f) in enumerate(feats.items())}
    sum((f * torch.from_numpy(ws[k]).to(DEV)).sum() for k, f in feats.items()).backward()
    torch.cuda.synchronize()
    # oracle: float64 autograd of FPN(backbone_maps(taps)) at the ORACLE's taps, then the encoder's backward for those d taps
    taps_np, _ = oracle.vit_forward(cfg, w, x)
    taps_t = [torch.from_numpy(t).double().requires_grad_(True) for t in taps_np]
    wt = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in _fpn_weights(m).items()}
    ref = fpn_forward_t(backbone_maps_t(taps_t, g, g), wt)
    sum((ref[k] * torch.from_numpy(ws[k]).double()).sum() for k in ref).backward()
    for k, f in feats.items():
        # train mode: both builds run the encoder on bf16 MFMA operands (mixed precision, DiTEncoder docstring)
        assert rel_l2(f.detach().cpu().numpy(), ref[k].detach().numpy()) < 2e-2, k
    for k, r in wt.items():
        got = dict(m.fpn.named_parameters())[k].grad
        assert got is not None and tuple(got.shape) == tuple(r.shape), k
        assert rel_l2(got.cpu().numpy(), r.grad.numpy()) < 3e-2, k
    _, enc_ref = train_reference(cfg, w, x, [t.grad.float().numpy() for t in taps_t], taps=m.backbone.layer_idxs)
    st = m.backbone.dit._flat_state
    got = {name: p.grad.detach().cpu().numpy() for name, p, _, _ in st.named}
    for short, hf in (("0.w1", "encoder.layer.0.intermediate.dense.weight"), ("2.wq", "encoder.layer.2.attention.attention.query.weight"),
                      ("patch_w", "embeddings.patch_embeddings.projection.weight"), ("1.lam2", "encoder.layer.1.lambda_2"),
                      ("pos", "embeddings.position_embeddings")):
        assert rel_l2(got[short], enc_ref[hf]) < 3e-2, short
    # frozen FPN, trainable encoder: gradients still reach the encoder (the graph used to be cut silently)
    for p in m.parameters():
        p.grad = None
    for p in m.fpn.parameters():
        p.requires_grad_(False)
    feats = m(torch.from_numpy(x).to(DEV))
    sum((f * torch.from_numpy(ws[k]).to(DEV)).sum() for k, f in feats.items()).backward()
    assert all(p.grad is None for p in m.fpn.parameters())
    got2 = st.named[4][1].grad
    assert got2 is not None and float(got2.abs().max()) > 0
    with pytest.raises(RuntimeError, match="second time"):
        sum((f * torch.from_numpy(ws[k]).to(DEV)).sum() for k, f in feats.items()).backward()


@pytest.mark.parametrize("case", ["frozen_encoder", "fp8_encoder", "one_fpn_parameter_frozen"])
def test_fpn_parameter_gradients_follow_their_own_requires_grad_flag(case):
    """The reference's commented option (ref dit_backbone.py:74-76: backbone frozen, FPN + head trained) and a single frozen FPN
    parameter: every parameter that still requires a gradient gets one, equal to the float64 oracle's, and a frozen one gets
    none (the flags used to be read one slot off, so the neighbour's flag decided)."""
    cfg = cfgs.vit_micro()
    cfg.drop_path_rate = 0.0
    cfg.taps = [1, 1, 2, 3]
    w = synth.synth_weights(cfg, 3)
    B, size, g = 2, 64, 4
    x = synth.synth_images(B, size, size, seed=5, kind="uniform")
    torch.manual_seed(0)
    m = DiTWithFPN(config=cfg, compute_dtype="fp8" if case == "fp8_encoder" else "f32")
    m.backbone.dit.load_numpy(w)
    with torch.no_grad():
        for p in m.fpn.parameters():
            if p.dim() == 1:
                p.normal_(0.0, 0.05)
    m = m.to(DEV)
    frozen = set()
    if case == "one_fpn_parameter_frozen":
        m.train()
        frozen = {"inner_blocks.1.0.weight"}
        m.fpn.inner_blocks[1][0].weight.requires_grad_(False)
    else:
        for p in m.backbone.parameters():
            p.requires_grad_(False)
        if case == "fp8_encoder":
            m.eval()                                          # an inference-only build behind a trainable FPN
            m.backbone.dit.calibrate_fp8(torch.from_numpy(x).to(DEV))
        else:
            m.backbone.eval()
    xt = torch.from_numpy(x).to(DEV)
    feats = m(xt)
    ws = {k: _rand(80 + i, *f.shape, scale=1.0 / np.sqrt(f.numel())) for i, (k, f) in enumerate(feats.items())}
    sum((f * torch.from_numpy(ws[k]).to(DEV)).sum() for k, f in feats.items()).backward()
    torch.cuda.synchronize()
    # the oracle differentiates the FPN at the taps the build itself produced (the encoder's own parity is tested elsewhere)
    with torch.no_grad():
        hs = m.backbone.dit(xt, taps=m.backbone.layer_idxs).hidden_states
    taps_t = [hs[i].detach().cpu().double() for i in m.backbone.layer_idxs]
    wt = {k: torch.from_numpy(v).double().requires_grad_(True) for k, v in _fpn_weights(m).items()}
    ref = fpn_forward_t(backbone_maps_t(taps_t, g, g), wt)
    sum((ref[k] * torch.from_numpy(ws[k]).double()).sum() for k in ref).backward()
    tol = 3e-2
    for k, r in wt.items():
        got = dict(m.fpn.named_parameters())[k].grad
        if k in frozen:
            assert got is None, k
            continue
        assert got is not None, f"{k}: no gradient although it requires one"
        assert rel_l2(got.cpu().numpy(), r.grad.numpy()) < tol, k
    if case == "one_fpn_parameter_frozen":
        assert m.backbone.dit._flat_state.named[4][1].grad is not None
    else:
        assert all(p.grad is None for p in m.backbone.parameters())


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_detector_input_transform_ragged_list(dtype):
    """ref model.py:45-55,87-88: List[Tensor[3, h, w]] in [0, 1] -> normalised 224 x 224 batch; boxes follow the resize."""
    sizes = [(300, 200), (224, 224), (97, 411), (640, 480)]
    imgs = [torch.from_numpy(np.clip(0.5 + 0.3 * _rand(10 + i, 3, h, w), 0, 1)).to(dtype) for i, (h, w) in enumerate(sizes)]
    boxes = [torch.tensor([[10.0, 20.0, 100.0, 150.0]]) for _ in sizes]
    t = DetectorInputTransform()
    il, targets = t([im.to(DEV) for im in imgs], [{"boxes": b.to(DEV), "labels": torch.ones(1)} for b in boxes])
    assert tuple(il.tensors.shape) == (4, 3, 224, 224) and il.tensors.dtype == torch.float32
    assert il.image_sizes == [(224, 224)] * 4
    for i, ((h, w), im) in enumerate(zip(sizes, imgs)):
        ref = F.interpolate(((im.float() - 0.5) / 0.5)[None], size=(224, 224), mode="bilinear", align_corners=False)[0]
        assert rel_l2(il.tensors[i].cpu().numpy(), ref.numpy()) < 2e-6, i
        want = boxes[i] * torch.tensor([224.0 / w, 224.0 / h, 224.0 / w, 224.0 / h])
        assert torch.allclose(targets[i]["boxes"].cpu(), want)
    back = t.postprocess([{"boxes": tg["boxes"].clone()} for tg in targets], il.image_sizes, sizes)
    for b0, b1 in zip(boxes, back):
        assert torch.allclose(b1["boxes"].cpu(), b0, atol=1e-4)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_input_transform_fused_into_the_patch_embedding_load(dtype):
    """SURVEY.md 8(f)-2 / ref model.py:50-54: ldit_embed_bf16_images builds the bf16 patch matrix straight from the ragged [0, 1]
    image list - normalise + bilinear resize inside the im2col pass.  EQUAL (torch.equal: one shared blend statement,
    csrc/image_blend.h) to DetectorInputTransform -> ldit_embed_bf16, and within the bf16 embedding's gate of the float64
    embedding of F.interpolate's pixels."""
    cfg = cfgs.vit_tiny()
    w = synth.synth_weights(cfg, 2)
    sizes = [(300, 200), (224, 224), (97, 411), (640, 480), (31, 17)]
    imgs = [torch.from_numpy(np.clip(0.5 + 0.3 * _rand(20 + i, 3, h, wd), 0, 1)).to(dtype).to(DEV) for i, (h, wd) in enumerate(sizes)]
    pw = torch.from_numpy(w["embeddings.patch_embeddings.projection.weight"]).to(DEV)
    pw16 = ops.cast_bf16(pw.reshape(cfg.hidden_size, -1).contiguous())
    pb = torch.from_numpy(w["embeddings.patch_embeddings.projection.bias"]).to(DEV)
    cls = torch.from_numpy(w["embeddings.cls_token"].reshape(-1)).to(DEV)
    pos = torch.from_numpy(w["embeddings.position_embeddings"].reshape(-1, cfg.hidden_size)).to(DEV)
    il, _ = DetectorInputTransform()(imgs)
    two_step = ops.embed_bf16(il.tensors, pw16, pb, cls, pos, 16)
    fused = ops.embed_bf16_images(imgs, pw16, pb, cls, pos, 16, size=224)
    assert torch.equal(fused, two_step)
    # against torch's own resize + a float64 embedding (bf16 operand rounding is all that separates them)
    ref_px = torch.stack([F.interpolate(((im.float().cpu() - 0.5) / 0.5)[None], size=(224, 224), mode="bilinear", align_corners=False)[0]
                          for im in imgs]).double()
    patches = ref_px.reshape(5, 3, 14, 16, 14, 16).permute(0, 2, 4, 1, 3, 5).reshape(5, 196, 768)
    emb = patches @ pw.cpu().double().reshape(cfg.hidden_size, -1).t() + pb.cpu().double() + pos.cpu().double()[1:]
    assert rel_l2(fused[:, 1:].cpu().numpy(), emb.numpy()) < 5e-3
    assert rel_l2(fused[:, 0].cpu().numpy(), (cls + pos[0]).cpu().double().expand(5, -1).numpy()) < 1e-6


@pytest.mark.parametrize("build", ["f32", "bf16", "f32x3", "fp8"])
def test_forward_image_list_equals_transform_then_forward(build):
    """ldit_vit_forward_images (DiTEncoder.forward_image_list / DetectorInputTransform.encode): the whole forward fed by the ragged
    list - every build returns the taps of the two-step path bit for bit (the low-precision builds without ever writing the fp32
    224 x 224 batch; the fp32 build, whose GEMM gathers pixels by LDS-DMA, produces it in its workspace)."""
    cfg = cfgs.vit_micro() if build != "fp8" else cfgs.vit_micro()
    size = 64
    w = synth.synth_weights(cfg, 4)
    sizes = [(80, 50), (64, 64), (33, 129)]
    imgs = [torch.from_numpy(np.clip(0.5 + 0.3 * _rand(30 + i, 3, h, wd), 0, 1)).to(DEV) for i, (h, wd) in enumerate(sizes)]
    m = DiTEncoder(cfg, compute_dtype=build).load_numpy(w).to(DEV).eval()
    t = DetectorInputTransform(fixed_size=(size, size))
    with torch.no_grad():
        batch = t(imgs)[0].tensors
        if build == "fp8":
            m.calibrate_fp8(batch)
        want = m(batch).hidden_states
        got = t.encode(m, imgs).hidden_states
        got16 = t.encode(m, [im.half() for im in imgs]).hidden_states
        want16 = m(t([im.half() for im in imgs])[0].tensors).hidden_states
    for tp in cfg.taps:
        assert torch.equal(got[tp], want[tp]), (build, tp)
        # fp16 images in -> fp16 taps out (as DiTEncoder.forward does for an fp16 batch): the fp32 taps of the two-step path, narrowed
        assert got16[tp].dtype == torch.float16 and torch.equal(got16[tp], ops.narrow_f16(want16[tp].contiguous())), (build, tp)
    m.train()
    with pytest.raises(RuntimeError, match="inference entry"):
        m.forward_image_list(imgs, size=(size, size))
    with pytest.raises(ValueError, match="channel dimension"):
        m.eval().forward_image_list([imgs[0][:2]], size=(size, size))


def test_encoder_accepts_fp16_pixels_without_a_host_cast():
    """ref trainer.py:153-155: the trainer hands fp16 images to the model; the encoder widens them with its own kernel and
    returns fp16 hidden states."""
    cfg = cfgs.vit_tiny()
    m = DiTEncoder(cfg).load_numpy(synth.synth_weights(cfg, 1)).to(DEV).eval()
    x = torch.from_numpy(synth.synth_images(2, 224, 224, seed=1234)).to(DEV)
    with torch.no_grad():
        h32 = m(x.half().float()).hidden_states
        h16 = m(x.half()).hidden_states
    for t in cfg.taps:
        assert h16[t].dtype == torch.float16
        assert torch.equal(h16[t], h32[t].half())
    with pytest.raises(ValueError, match="float32 or float16"):
        m(x.to(torch.bfloat16))


@pytest.mark.parametrize("gh,gw,scale", [(14, 14, 4.0), (14, 14, 2.0), (14, 14, 0.5), (6, 4, 4.0), (5, 7, 0.5), (3, 3, 2.0)])
def test_tap_to_map_adjoint(gh, gw, scale):
    """Backward of the tap post-processing (ref dit_backbone.py:50-61 under loss.backward()): the gather-form adjoint kernel
    against torch autograd of the same F.interpolate call."""
    B, Cc = 2, 64
    tap = _rand(30, B, gh * gw + 1, Cc)
    w = _rand(31, B, Cc, int(gh * scale), int(gw * scale))
    t = torch.from_numpy(tap).to(DEV).requires_grad_(True)
    m = ops.tap_to_map_autograd(t, gh, gw, scale)
    (m * torch.from_numpy(w).to(DEV)).sum().backward()
    r = torch.from_numpy(tap).double().requires_grad_(True)
    rm = F.interpolate(r[:, 1:, :].permute(0, 2, 1).reshape(B, Cc, gh, gw), scale_factor=scale, mode="bilinear", align_corners=False)
    (rm * torch.from_numpy(w).double()).sum().backward()
    assert rel_l2(m.detach().cpu().numpy(), rm.detach().numpy()) < 1e-6
    assert rel_l2(t.grad.cpu().numpy(), r.grad.numpy()) < 1e-6
    assert float(t.grad[:, 0].abs().max()) == 0.0                 # the CLS token is sliced away


def test_backbone_train_mode_is_differentiable():
    """DiTBackbone.forward under autograd (the reference trains through it): gradients flow from p2..p5 through the rescale
    adjoint into the encoder's backward; compared with the float64 autograd oracle of the whole chain."""
    from layoutdit_amd.modeling import DiTBackbone
    from oracle.vit_oracle_torch import train_reference
    cfg = cfgs.vit_micro()
    cfg.drop_path_rate = 0.0
    cfg.taps = [1, 1, 2, 3]
    w = synth.synth_weights(cfg, 3)
    x = synth.synth_images(2, 64, 64, seed=5, kind="uniform")
    bb = DiTBackbone(config=cfg, compute_dtype="bf16")
    bb.dit.load_numpy(w)
    bb = bb.to(DEV).train()
    feats = bb(torch.from_numpy(x).to(DEV))
    ws = [torch.from_numpy(_rand(40 + i, *f.shape, scale=1.0 / np.sqrt(f.numel()))).to(DEV) for i, f in enumerate(feats.values())]
    sum((f * wi).sum() for f, wi in zip(feats.values(), ws)).backward()
    torch.cuda.synchronize()
    # oracle: upstream gradient at each tap = adjoint of the reference's own post-processing, by torch autograd on the CPU
    g = 4
    dtaps = []
    for (name, f), wi, s in zip(feats.items(), ws, bb.scales):
        t = torch.zeros(2, g * g + 1, cfg.hidden_size, dtype=torch.float64, requires_grad=True)
        m = t[:, 1:, :].permute(0, 2, 1).reshape(2, cfg.hidden_size, g, g)
        if s != 1.0:
            m = F.interpolate(m, scale_factor=s, mode="bilinear", align_corners=False)
        (m * wi.cpu().double()).sum().backward()
        dtaps.append(t.grad.float().numpy())
    _, ref = train_reference(cfg, w, x, dtaps, taps=bb.layer_idxs)
    st = bb.dit._flat_state
    got = {name: p.grad.detach().cpu().numpy() for name, p, _, _ in st.named}
    for short, hf in (("0.w1", "encoder.layer.0.intermediate.dense.weight"), ("2.wq", "encoder.layer.2.attention.attention.query.weight"),
                      ("patch_w", "embeddings.patch_embeddings.projection.weight"), ("1.lam2", "encoder.layer.1.lambda_2")):
        assert rel_l2(got[short], ref[hf]) < 3e-2, short
