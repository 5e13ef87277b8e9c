"""The real-weights parity hook (tests/parity_checkpoint.py, launched by scripts/parity_checkpoint.py): on the CPU its loader -
safe reading of a local checkpoint in each accepted key layout, geometry from the tensor shapes - against a synthetic checkpoint
written in the transformers-4.49 layout the reference pins; on the GPU the whole report on that checkpoint."""
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip("torch")

from layoutdit_amd import config as cfgs, synth                      # noqa: E402
from layoutdit_amd.modeling.keys import to_v5                        # noqa: E402
from tests import parity_checkpoint as pc                            # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(tmp_path, name, sd, wrap=None):
    path = str(tmp_path / name)
    obj = {k: torch.from_numpy(v.copy()) for k, v in sd.items()}
    torch.save({wrap: obj} if wrap else obj, path)
    return path


@pytest.mark.parametrize("layout", ["v4", "v5", "detector_prefixed", "wrapped_fp16", "safetensors"])
def test_loader_reads_every_accepted_layout(tmp_path, layout):
    cfg = cfgs.vit_micro()
    w = synth.synth_weights(cfg, 5)
    if layout == "v4":
        path = _write(tmp_path, "m.bin", w)
    elif layout == "v5":
        path = _write(tmp_path, "m.pt", {to_v5(k): v for k, v in w.items()})
    elif layout == "detector_prefixed":                # what ref model.py:90-121 writes: the whole detector's state dict
        sd = {"model.backbone.backbone.dit." + k: v for k, v in w.items()}
        sd["model.rpn.head.conv.0.0.weight"] = np.zeros((4, 4, 3, 3), np.float32)
        path = _write(tmp_path, "epoch_10_cpu.pth", sd)
    elif layout == "wrapped_fp16":
        path = _write(tmp_path, "m.pth", {k: v.astype(np.float16) for k, v in w.items() if "mask_token" not in k and "pooler" not in k},
                      wrap="state_dict")
    else:
        from safetensors.numpy import save_file
        path = str(tmp_path / "model.safetensors")
        save_file({k: v for k, v in w.items()}, path)
    got_cfg, got_w, keys = pc.load_checkpoint(path)
    assert (got_cfg.hidden_size, got_cfg.num_hidden_layers, got_cfg.intermediate_size, got_cfg.patch_size, got_cfg.image_size) == \
        (cfg.hidden_size, cfg.num_hidden_layers, cfg.intermediate_size, cfg.patch_size, cfg.image_size)
    assert got_cfg.num_attention_heads == cfg.hidden_size // 64 and got_cfg.taps == cfg.taps
    assert set(got_w) == set(synth.param_shapes(cfg))
    for k, v in w.items():
        if layout == "wrapped_fp16":
            if "mask_token" in k or "pooler" in k:
                assert k in keys["missing"] and not got_w[k].any()
            else:
                np.testing.assert_array_equal(got_w[k], v.astype(np.float16).astype(np.float32))
        else:
            np.testing.assert_array_equal(got_w[k], v)
    if layout == "detector_prefixed":
        assert keys["unexpected"] == ["model.rpn.head.conv.0.0.weight"]


def test_loader_refuses_what_is_not_this_encoder(tmp_path):
    cfg = cfgs.vit_micro()
    w = synth.synth_weights(cfg, 5)
    bad = dict(w)
    del bad["encoder.layer.1.attention.output.dense.weight"]
    with pytest.raises(SystemExit, match="lacks"):
        pc.load_checkpoint(_write(tmp_path, "a.bin", bad))
    rel = dict(w)
    rel["encoder.layer.0.attention.attention.relative_position_bias.relative_position_bias_table"] = np.zeros((9, 2), np.float32)
    with pytest.raises(SystemExit, match="relative-position"):
        pc.load_checkpoint(_write(tmp_path, "b.bin", rel))
    with pytest.raises(SystemExit, match="not a BEiT"):
        pc.load_checkpoint(_write(tmp_path, "c.bin", {"fc.weight": np.zeros((2, 2), np.float32)}))
    # a pickle that is not a plain tensor dict is refused by the safe loader itself, not executed
    import pickle
    evil = str(tmp_path / "evil.bin")
    with open(evil, "wb") as f:
        pickle.dump({"x": os.system}, f)
    with pytest.raises(Exception):
        pc.load_checkpoint(evil)


def test_compare_and_gates():
    ref = np.array([[1.0, -2.0, 0.5, 100.0]])
    r = pc.compare(ref * (1 + 1e-4), ref)
    assert abs(r["rel_l2"] - 1e-4) < 1e-9 and abs(r["worst_elem"] - 1e-4) < 1e-9 and r["cosine"] > 0.999999
    assert pc.verdict("f32", [r]) and pc.verdict("bf16", [r]) and pc.verdict("fp8", [r])
    r2 = pc.compare(ref * 1.05, ref)
    assert not pc.verdict("f32x3", [r2]) and not pc.verdict("bf16", [r2]) and pc.verdict("fp8", [r2])


@pytest.mark.gpu
def test_report_on_a_synthetic_checkpoint_written_in_the_4_49_layout(tmp_path):
    cfg = cfgs.vit_micro()
    path = _write(tmp_path, "pytorch_model.bin", synth.synth_weights(cfg, 5))
    out = str(tmp_path / "report.json")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_checkpoint.py"), "--checkpoint", path, "--json", out],
                         capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0, res.stdout[-3000:] + res.stderr[-3000:]
    import json
    rep = json.load(open(out))
    assert rep["geometry"]["hidden"] == cfg.hidden_size and rep["taps"] == cfg.taps
    for b in ("f32", "f32x6", "f32x3", "bf16", "fp8"):
        assert rep["builds"][b]["pass"], (b, rep["builds"][b])
    assert max(r["rel_l2"] for r in rep["builds"]["f32"]["per_tap"]) < 2e-5
    # another input size: the bicubically resampled position table on both sides
    res = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "parity_checkpoint.py"), "--checkpoint", path, "--size", "96",
                          "--builds", "f32,bf16"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert res.returncode == 0 and res.stdout.count("PASS") == 2, res.stdout[-3000:] + res.stderr[-3000:]
