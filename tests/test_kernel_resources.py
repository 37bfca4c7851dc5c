"""Build-time properties of the hand-allocated kernel (no GPU needed: hipcc cross-compiles gfx950).

csrc/tools/kernels_tdnn_chain4.hip (developer build only) keeps its 256 accumulator registers in AGPRs that the COMPILER does not know about (every access is inline
assembly naming the registers).  That is only sound while hipcc itself puts nothing there: no spills (it parks spilled VGPRs in
free AGPRs first) and no AGPR operand outside the assembly blocks."""
import os
import re
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(REPO, "asv-subtools_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_chain4_kernel_has_no_spills_and_no_compiler_owned_agprs(tmp_path):
    out = tmp_path / "chain4.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-Xclang", "-target-feature", "-Xclang",
           "-packed-fp32-ops", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", str(out), "-DASV_WITH_ABLATION", os.path.join(CSRC, "tools", "kernels_tdnn_chain4.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    kernels = re.findall(r"Function Name: (\S+)", r.stderr)
    spills = [int(x) for x in re.findall(r"VGPRs Spill: (\d+)", r.stderr)]
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    agprs = [int(x) for x in re.findall(r"AGPRs: (\d+)", r.stderr)]
    assert len(kernels) >= 2 and len(spills) == len(kernels)
    assert all(s == 0 for s in spills) and all(s == 0 for s in scratch), (spills, scratch)
    assert all(a == 256 for a in agprs), agprs                   # the kernel descriptor covers the whole accumulator file
    in_asm, bad, n_mfma = False, [], 0
    for line in out.read_text().splitlines():
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        code = line.split(";")[0]
        if "v_mfma" in code:
            n_mfma += 1
        if not in_asm and re.search(r"\ba\[?\d", code) and re.match(r"\s+[a-z]", code):
            bad.append(line.strip())
    assert n_mfma > 1000
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_persistent_f32x_convolution_kernels_keep_their_weights_in_registers(tmp_path):
    """grid_conv_x3_pers32_kernel / _pers64_kernel (kernels_conv2d_x3.hip) hold all 36 hi / lo weight fragments of their layer (144
    registers) for a whole run of tiles and are launched two workgroups per CU: that design stands only while the compiler keeps the
    kernels inside 256 registers WITHOUT spilling into scratch (a spilled fragment would be re-read from memory inside the K loop)
    and the LDS rings leave room for two workgroups (the 64-row / 8-wave form: one)."""
    out = tmp_path / "x3.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-Xclang", "-target-feature", "-Xclang",
           "-packed-fp32-ops", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", str(out), os.path.join(CSRC, "kernels_conv2d_x3.hip")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    seen = 0
    for b in blocks:
        name = b.split()[0]
        if "pers32_kernel" not in name and "pers64_kernel" not in name:
            continue
        seen += 1
        vgprs = int(re.search(r"VGPRs: (\d+)", b).group(1))
        scratch = int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1))
        lds = int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1))
        assert vgprs <= 256 and scratch == 0, (name, vgprs, scratch)
        one_per_cu = "pers64_kernelILi1ELi2E" in name or "pers64_kernelILi2ELi2E" in name          # WM = 2: 8 waves, one workgroup per CU
        assert lds * (1 if one_per_cu else 2) <= 163840, (name, lds)
    assert seen >= 8, seen            # pers32 x 2 split types, pers64 x 2 split types x {32-row, 32-row pipelined, 64-row}


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
@pytest.mark.parametrize("src,extra", [("kernels_tdnn_p8.hip", []), ("kernels_tdnn_p8x.hip", ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"])])
def test_8phase_kernels_have_no_spills_and_no_compiler_visible_vector_loads(tmp_path, src, extra):
    """The 8-phase kernels (kernels_tdnn_p8.hip, kernels_tdnn_p8x.hip) issue every vector-memory read as LDS-DMA from inline assembly
    and wait with counted `s_waitcnt vmcnt(N)`: hipcc counts only the vector-memory operations it knows of, so ONE load of its own in
    the tile loop - or one spilled register (scratch is vector memory) - turns into waits that, in hardware terms, drain the DMA pieces
    in flight (the first p8x build spilled 30 registers across the K loop and got a `vmcnt(0)` at the top of every tile).  Build-time
    check: no spills, no scratch, 2 waves per SIMD, and no global / scratch / buffer load instruction of the compiler's behind a kernel's first DMA."""
    out = tmp_path / "k.s"
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-Wno-inline-asm"] + extra + [
        "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", str(out), os.path.join(CSRC, src)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    assert len(blocks) >= 4
    for b in blocks:
        name = b.split()[0]
        assert int(re.search(r"VGPRs Spill: (\d+)", b).group(1)) == 0, name
        assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) == 0, name
        assert int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1)) == 2, name
    in_asm, bad, n_dma, dma_seen, checked = False, [], 0, False, False
    for line in out.read_text().splitlines():
        if re.match(r"_Z\w+:", line):                          # a kernel begins (the persistent production forms are the ones checked:
            dma_seen = False                                   #  the one-tile development forms of kernels_tdnn_p8.hip keep plain loads)
            checked = "p8p_kernel" in line or "p8x_kernel" in line
        if "#ASMSTART" in line:
            in_asm = True
            continue
        if "#ASMEND" in line:
            in_asm = False
            continue
        code = line.split(";")[0].strip()
        if in_asm and code.startswith("global_load_lds_dwordx4"):
            n_dma += 1
            dma_seen = True
        # (in front of a kernel's first DMA a load of hipcc's own is harmless: the tap table, which it fetches from the kernarg segment
        #  with a vector load when the kernel selects an entry by lane)
        if not in_asm and dma_seen and checked and re.match(r"(global_load|scratch_load|scratch_store|buffer_load|flat_load)", code):
            bad.append(code)
    assert n_dma >= 40
    assert not bad, bad[:5]


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_res2_chain_kernel_forms_fit_their_lds_and_registers(tmp_path):
    """res2_chain_kernel<ET, 6 | 7> (kernels_res2.hip): the weight fragments of a branch (96 registers) + 3 / 4 accumulators live in
    registers and the fragment loads are inline assembly with counted waits - a spill would put compiler-tracked vector memory next to
    them (the first FR = 7 build spilled the 7 DMA source addresses of a lane: scratch reloads between the window pieces).  The images
    B and X share one buffer: 111 104 / 127 488 bytes of LDS (an array of registers in the streaming step was once promoted to LDS by
    hipcc: + 32 KiB, silently)."""
    r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-c",
                        "-Rpass-analysis=kernel-resource-usage", "-o", str(tmp_path / "res2.o"), os.path.join(CSRC, "kernels_res2.hip")], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]
    assert len(blocks) == 4
    for b in blocks:
        name = b.split()[0]
        fr = 7 if "ELi7E" in name else 6
        assert ("ELi%dE" % fr) in name
        assert int(re.search(r"VGPRs Spill: (\d+)", b).group(1)) == 0, name
        assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) == 0, name
        assert int(re.search(r"Occupancy \[waves/SIMD\]: (\d+)", b).group(1)) == 2, name
        assert int(re.search(r"LDS Size \[bytes/block\]: (\d+)", b).group(1)) == {6: 111104, 7: 127488}[fr], name


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="no hipcc")
def test_f32m_kernels_fit_their_registers_and_lds(tmp_path):
    """The 8-bit correction kernels of the f32m form (round 6): tdnn_chainm_kernel - 64 accumulators, two sets of weight fragments, one
    workgroup of 8 waves per CU = 256 registers per lane - must not spill at all (its K loops are bound by the weight stream: a spilled
    fragment would add to it) and must fit one CU's LDS; tdnn_gemm_x3m_kernel<false> - 128 accumulators at two workgroups per CU - may park
    a few lane-constant values outside its K loop (pinned here at <= 8 spilled registers), with LDS for two workgroups."""
    seen = {}
    for name in ("kernels_tdnn_chainm.hip", "kernels_tdnn_x3m.hip"):
        out = tmp_path / (name + ".s")
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(REPO, "include"), "-I" + CSRC, "-Xclang", "-target-feature", "-Xclang",
               "-packed-fp32-ops", "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-o", str(out), os.path.join(CSRC, name)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        for block in re.split(r"remark: [^\n]*Function Name: ", r.stderr)[1:]:
            fn = block.split()[0]
            seen[fn] = {k: int(re.search(p, block).group(1)) for k, p in (("vgprs", r" VGPRs: (\d+)"), ("spill", r"VGPRs Spill: (\d+)"),
                                                                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("lds", r"LDS Size \[bytes/block\]: (\d+)"))}
    chain = [v for k, v in seen.items() if "tdnn_chainm_kernel" in k]                 # <IMG, DEV>: f32 / image rows in, production / developer aids
    x3m = [v for k, v in seen.items() if "tdnn_gemm_x3m_kernel" in k]                 # <GENERIC, XIMG>
    x3m_img = [v for k, v in seen.items() if "tdnn_gemm_x3m_kernelILb0ELb1" in k or "tdnn_gemm_x3m_kernelILb1ELb1" in k]
    assert len(chain) == 4 and len(x3m) == 4 and len(x3m_img) == 2, sorted(seen)
    for c in chain:
        assert c["spill"] == 0 and c["scratch"] == 0 and c["vgprs"] <= 256 and c["lds"] <= 160 * 1024, chain
    for v in x3m_img:                                                                 # image rows in: no conversion pass, nothing parked
        assert v["spill"] == 0, x3m_img
    for v in x3m:
        assert v["spill"] <= 8 and v["vgprs"] <= 256 and 2 * v["lds"] <= 160 * 1024, x3m
