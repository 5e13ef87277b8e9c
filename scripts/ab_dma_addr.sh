for i in 1 2; do
echo "== sbase (new) =="; timeout -k 10 300 python scripts/gemm_vendor_ref.py 2>&1 | grep -v amdgpu.ids | cut -c1-78
echo "== vaddr (old) =="; timeout -k 10 300 python -c "
import sys; sys.argv=['x']
import layoutdit_amd._lib as L; L.LIB_PATH='layoutdit_amd/csrc/build/libldit_vaddr.so'
import runpy; runpy.run_path('scripts/gemm_vendor_ref.py', run_name='__main__')" 2>&1 | grep -v amdgpu.ids | cut -c1-78
done
