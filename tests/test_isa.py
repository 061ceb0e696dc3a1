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
