"""Build-time guard: no kernel of the GEMM / convolution file may spill registers to scratch memory.
(A runtime-indexed staging array once went to scratch silently: the pointwise layers wrote 2-4x their output
bytes to HBM and the whole forward lost 6 %.)  Compiles conv_igemm.hip with the resource-usage remarks on."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
@pytest.mark.parametrize("src", ["conv_igemm.hip", "conv_wino.hip", "mano_lbs.hip", "stem_pool.hip"])
def test_hot_kernels_do_not_spill(tmp_path, src):
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/hands_amd/csrc",
                        "-fno-fast-math", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "-c",
                        os.path.join(ROOT, "hands_amd", "csrc", src), "-o", str(tmp_path / "o.o")] +
                       (["-fno-slp-vectorize"] if src == "conv_wino.hip" else []),      # as hands_amd/csrc/Makefile builds it
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    if src == "mano_lbs.hip":
        # the fused MANO kernel must keep FOUR workgroups per CU (<= 128 registers, <= 40 KB of LDS): tools/prof_mano.py measured
        # a second dispatch round 25 us late when an instrumented copy needed 130
        b = next(b for b in re.split(r"Function Name: ", p.stderr)[1:] if "mano_heads_kernelILb0" in b.split()[0])
        vg, lds = int(re.search(r"VGPRs: (\d+)", b).group(1)), int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1))
        assert vg <= 128 and lds <= 40960, (vg, lds)
    if src == "conv_wino.hip":      # three workgroups per CU: <= 168 registers, <= 53 KB of LDS, accumulators never copied to AGPRs
        for b in re.split(r"Function Name: ", p.stderr)[1:]:
            vg, ag = int(re.search(r"VGPRs: (\d+)", b).group(1)), int(re.search(r"AGPRs: (\d+)", b).group(1))
            lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1))
            assert vg + ag <= 168 and 40960 < lds <= 54272, (b.split()[0], vg, ag, lds)
    names = re.findall(r"Function Name: (\S+)", p.stderr)
    scratch = [int(v) for v in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", p.stderr)]
    spills = [int(v) for v in re.findall(r"VGPRs Spill: (\d+)", p.stderr)]
    assert names and len(names) == len(scratch) == len(spills)
    bad = [(n, s, v) for n, s, v in zip(names, scratch, spills) if s or v]
    assert not bad, bad


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_fp32_conv_kernels_keep_four_workgroups_per_cu(tmp_path):
    """Every exact-fp32 instantiation of the plain convolution kernel must fit 4 workgroups of 4 waves on a CU:
    <= 128 VGPRs (512 per SIMD lane / 4 waves) and <= 40 KB of LDS (160 KB / 4).  The 256x64 tile once sat at
    130 VGPRs / 51 KB = 3 per CU."""
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/hands_amd/csrc",
                        "-fno-fast-math", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "-c",
                        os.path.join(ROOT, "hands_amd", "csrc", "conv_igemm.hip"), "-o", str(tmp_path / "o.o")],
                       capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    blocks = re.split(r"Function Name: ", p.stderr)[1:]
    seen = blocked = 0
    for b in blocks:
        name = b.split()[0]
        m = re.search(r"conv_igemm_f32_kernelILi(\d)ELi(\d)ELi(\d)ELi(\d)ELb(\d)ELi(\d+)E", name)
        if not m or m.group(4) != "0":     # plain kernel, PREC 0 (exact fp32)
            continue
        if m.group(6) != "0":      # blocked summation (HANDS_SUM_BLOCK128 / 64): a second accumulator set, two workgroups per CU
            vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1)) + int(re.search(r"AGPRs: (\d+)", b).group(1))
            assert vgprs <= 256 and int(re.search(r"VGPRs Spill: (\d+)", b).group(1)) == 0, (name, vgprs)
            blocked += 1
            continue
        if m.group(5) == "1":      # PRE (BatchNorm -> LeakyReLU on the operand, handoccnet_light's pre-activation units only):
            continue               # two more staging vectors, 134 VGPRs = 3 workgroups per CU, measured +0.6-1 % over the separate launch
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        agprs = int(re.search(r"AGPRs: (\d+)", b).group(1))
        lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1))
        assert vgprs + agprs <= 128 and lds <= 40960, (name, vgprs, agprs, lds)
        seen += 1
    assert seen == 6 and blocked == 12


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not available")
def test_vit_attention_fits_two_workgroups_per_cu(tmp_path):
    """attention_kernel<12, 80> (csrc/transformer.hip): 12 waves per workgroup (a multiple of the 4 SIMDs: three each), <= 80
    registers (six waves per SIMD) and <= 80 KB of dynamic LDS, i.e. TWO workgroups per CU, no scratch beyond a handful of
    spilled registers.  The 6-wave / 168-register form of rounds 1-3 fitted one workgroup per CU with a 2-2-1-1 SIMD load
    (tools/prof_attn.py) and ran 35 % slower."""
    src = os.path.join(ROOT, "hands_amd", "csrc", "transformer.hip")
    p = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", f"-I{ROOT}/include", f"-I{ROOT}/hands_amd/csrc",
                        "-fno-fast-math", "-ffp-contract=off", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o",
                        str(tmp_path / "o.o")], capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-2000:]
    b = next(b for b in re.split(r"Function Name: ", p.stderr)[1:] if "attention_kernelILi12ELi80" in b.split()[0])
    vg = int(re.search(r"VGPRs: (\d+)", b).group(1))
    spill = int(re.search(r"VGPRs Spill: (\d+)", b).group(1))
    occ = int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1))
    assert vg <= 80 and occ >= 6 and spill <= 16, (vg, occ, spill)
    text = open(src).read()
    assert "dim3(768)" in text and "attention_kernel<12, 80>" in text          # 12 waves
    m = re.search(r"constexpr int attention_lds_bytes\(\) \{\s*return 4 \* \((.*?)\);", text, re.S)
    assert m, "attention_lds_bytes() not found"
    TW, D = 12, 80
    assert 2 * 4 * max(16 * TW * (D + 4), D * (16 * TW + 4)) <= 160 * 1024       # two workgroups' LDS fit a CU
