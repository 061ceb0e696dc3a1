"""ISA-level invariants of the tile kernels that the source cannot express.

glds16 (csrc/transform.hip) writes M0 inside inline assembly: `s_mov_b32 m0, <lds base>` in front of a
`global_load_lds_dwordx4`. hipcc treats M0 as a reserved register, so the write cannot be declared as a clobber (it warns and
ignores it); what keeps this correct is that NOTHING else in these kernels uses M0 -- no `s_movrel` / `v_movrel` indirect
register indexing, no `s_sendmsg`, no GWS / LDS-direct instruction of the compiler's own. This test pins that on the ISA the
product is built from, so a compiler upgrade that starts to keep a value in M0 across the asm fails here instead of silently
corrupting a tile."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "raht-3dgs-codec_amd", "csrc")
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("tu", ["transform.hip", "transform_mx.hip"])
def test_m0_is_only_touched_by_the_lds_direct_loads(tmp_path, tu):
    out = tmp_path / "transform.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-fno-fast-math",
                        "-ffp-contract=on", "-S", "--offload-device-only", os.path.join(CSRC, tu), "-o", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = out.read_text().splitlines()
    code = [ln.split(";")[0].strip() for ln in lines]
    uses = [i for i, ln in enumerate(code) if re.search(r"\bm0\b", ln) and not ln.startswith(".")]
    assert len(uses) >= 8, "expected the LDS-direct loads of the tile kernels"
    n_glds = sum(1 for ln in code if ln.startswith("global_load_lds_dwordx4"))
    assert n_glds == len(uses), (n_glds, len(uses))
    for i in uses:
        assert re.fullmatch(r"s_mov_b32 m0, s\d+", code[i]), f"unexpected use of m0: {lines[i]!r}"
        nxt = [ln for ln in code[i + 1:i + 4] if ln]
        assert nxt[0].startswith("s_nop") and nxt[1].startswith("global_load_lds_dwordx4"), (lines[i], nxt)
    # instructions that read M0 implicitly must not appear at all in this translation unit
    for ln in code:
        assert not re.match(r"(s_movrel|v_movrel|s_sendmsg|ds_gws|s_set_gpr_idx|v_interp)", ln), ln


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("tu,at_least", [("transform.hip", 16), ("transform_mx.hip", 10)])
def test_tile_kernels_do_not_spill(tu, at_least):
    """Any scratch use in a tile / top kernel is a regression (a kernel that touches scratch lost 30 %)."""
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "reg_report.sh"), tu], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    rows = [ln for ln in r.stdout.splitlines() if "vgpr" in ln]
    assert len(rows) >= at_least, r.stdout[-2000:]
    for ln in rows:
        m = re.search(r"spill (\d+) scratch (\d+)", ln)
        assert m and int(m.group(1)) == 0 and int(m.group(2)) == 0, ln


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_plan_kernels_save_and_restore_m0_around_their_writelanes(tmp_path):
    """level_extent_kernel collects its per-level words with `v_writelane_b32 v, s, m0` in inline assembly (no writelane builtin
    in this hipcc; two SGPR operands exceed the constant bus, so the lane select goes through M0). The compiler does not track M0
    as a clobber: the asm saves and restores it. Pinned here: every use of M0 in plan.hip is one of those four instructions."""
    out = tmp_path / "plan.s"
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-fno-fast-math",
                        "-ffp-contract=on", "-S", "--offload-device-only", os.path.join(CSRC, "plan.hip"), "-o", str(out)],
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    code = [ln.split(";")[0].strip() for ln in out.read_text().splitlines()]
    uses = [ln for ln in code if re.search(r"\bm0\b", ln) and not ln.startswith(".")]
    assert sum(1 for ln in uses if ln.startswith("v_writelane_b32")) >= 3, uses[:10]
    for ln in uses:
        assert re.fullmatch(r"s_mov_b32 s\d+, m0|s_mov_b32 m0, s\d+|v_writelane_b32 v\d+, s\d+, m0", ln), f"unexpected use of m0: {ln!r}"
    saves = sum(1 for ln in uses if re.fullmatch(r"s_mov_b32 s\d+, m0", ln))
    sets = sum(1 for ln in uses if re.fullmatch(r"s_mov_b32 m0, s\d+", ln))
    assert saves >= 1 and sets == 2 * saves, (saves, sets)              # (set for the writes, set back)


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
@pytest.mark.parametrize("tu", ["rlgr_seg.hip", "plan.hip"])
def test_coder_and_plan_kernels_do_not_spill(tu):
    """the segmented coder's kernels must keep 7-8 waves per SIMD (that is where their batched speed comes from): no scratch, at most
    72 VGPRs; the plan kernels: no scratch"""
    r = subprocess.run([HIPCC, "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"), "-fno-fast-math",
                        "-ffp-contract=on", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, tu), "-o", os.devnull],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    names = re.findall(r"Function Name: (\S+)", r.stderr)
    scratch = [int(x) for x in re.findall(r"ScratchSize \[bytes/lane\]: (\d+)", r.stderr)]
    vgprs = [int(x) for x in re.findall(r" VGPRs: (\d+)", r.stderr)]
    assert len(names) >= 8 and len(names) == len(scratch) == len(vgprs)
    for n, sc, v in zip(names, scratch, vgprs):
        assert sc == 0, (n, sc)
        if tu == "rlgr_seg.hip" and ("seg_encode" in n or "seg_decode" in n):
            assert v <= 72, (n, v)
