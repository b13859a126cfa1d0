"""CPU suite: the register and LDS budgets the kernels' residency rests on, read from the code objects the build made (llvm-readelf --notes on
the gfx950 part of build/product/*.o).  A toolchain change that pushes a kernel over its budget would be correct and slow -- e.g. the
headline's kernel, mzd_lds_kernel<8,false,4>, sits at exactly 256 registers (two wavefronts a SIMD) without scratch -- and no parity test
would see it."""
import os
import re
import subprocess

import pytest

import fuse_zstd_amd as mzd

LLVM = "/opt/rocm/lib/llvm/bin"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.skipif(not os.path.exists(LLVM + "/llvm-readelf"), reason="no ROCm LLVM tools on this machine")


def _kernels(obj, tmp_path):
    """{mangled kernel name: {field: int}} of the gfx950 code object inside a host object of the build."""
    fat, co = str(tmp_path / "fat.bin"), str(tmp_path / "k.co")
    subprocess.check_call([LLVM + "/llvm-objcopy", "--dump-section", ".hip_fatbin=" + fat, obj])
    subprocess.check_call([LLVM + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat, "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co])
    notes = subprocess.check_output([LLVM + "/llvm-readelf", "--notes", co], text=True)
    out, cur = {}, None
    for blk in notes.split("- .agpr_count:")[1:]:  # one block per kernel (amdhsa.kernels entries start with .agpr_count)
        f = dict((m.group(1), m.group(2)) for m in re.finditer(r"\.(\w+):\s+(\S+)", blk))
        out[f["name"]] = {k: int(v) for k, v in f.items() if v.isdigit()}
    return out


def test_the_kernels_keep_their_register_and_lds_budgets(tmp_path):
    mzd.build()
    b = os.path.join(ROOT, "fuse_zstd_amd", "csrc", "build", "product")
    small = _kernels(os.path.join(b, "mzd_lds.o"), tmp_path)
    def lds_kernel(g, d, xg, nw=1, nd=1):
        return small["_ZN3mzd2lw14mzd_lds_kernelILi%dELb%dELi%dELi%dELi%dEEEvNS_7LdsArgsE" % (g, d, xg, nw, nd)]
    # the headline (cfg4: 10 000 x 4 KiB in ONE round of five wavefronts a CU): 256 registers, nothing in scratch
    k = lds_kernel(8, 0, 4)
    assert k["vgpr_count"] <= 256 and k["vgpr_spill_count"] == 0 and k["private_segment_fixed_size"] == 0, k
    # 8/8 and 16/16 (many small files) likewise; the sustained shape 4/4 runs nine wavefronts a CU: 168 registers
    for g in (8, 16):
        k = lds_kernel(g, 0, g)
        assert k["vgpr_count"] <= 256 and k["private_segment_fixed_size"] == 0, (g, k)
    assert lds_kernel(4, 0, 4)["vgpr_count"] <= 168
    # one dictionary image a workgroup of five / eight decoding wavefronts (cfg5): two wavefronts a SIMD
    for nd in (5, 8):
        assert lds_kernel(8, 1, 8, 1, nd)["vgpr_count"] <= 256
    # the block pipeline: four groups of four wavefronts a CU -- 128 registers, an LDS image of at most 40 KiB (1 280-byte granules)
    blk = _kernels(os.path.join(b, "mzd_kernels.o"), tmp_path)
    for name in ("_ZN3mzd23mzd_decode_kernel_filesENS_10KernelArgsE", "_ZN3mzd23mzd_decode_kernel_tasksENS_10KernelArgsE"):
        k = blk[name]
        assert k["vgpr_count"] <= 128 and k["group_segment_fixed_size"] <= 40960 and k["max_flat_workgroup_size"] == 256, (name, k)
    # ... and the two-files-a-workgroup build of driver 1: the same registers, its images in the dynamic segment
    prs = _kernels(os.path.join(b, "mzd_kernels_pairs.o"), tmp_path)
    k = prs["_ZN3mzd23mzd_decode_kernel_pairsENS_10KernelArgsE"]
    assert k["vgpr_count"] <= 128 and k["group_segment_fixed_size"] == 0 and k["max_flat_workgroup_size"] == 512, k
