"""The product library's ISA must not contain VGPR spill stores / reloads that run before a block's exec restore
(tools/spill_lint.py: a hipcc placement bug that once corrupted observer_kernel<double>)."""
import importlib.util
import os
import shutil

import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_no_masked_spill_stores(tmp_path):
    spec = importlib.util.spec_from_file_location("spill_lint", os.path.join(ROOT, "tools", "spill_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    asm = mod.compile_asm(str(tmp_path / "wbc.s"))
    bad = mod.lint(asm)
    assert not bad, bad[:5]


# Kernels whose speed depends on a register budget: no scratch at all, at most this many registers (arch + accumulation).
# A change elsewhere in a shared body once pushed the observer-on fused tick into a 300-byte spill (+2.5 us per tick) and
# no functional test noticed.
BUDGET = [
    (r"fused_tick_kernelI[df]", 256),
    (r"dyn_sweep_kernelIdLi1ELi256", 256), (r"dyn_sweep_kernelIdLi11ELi256", 256), (r"dyn_sweep_kernelIfLi(1|11)ELi256", 256),
    (r"observer_kernelI[df]", 256),
    (r"qp_lane_kernelI[df]", 256),
    (r"qp_tile_kernelI[df]Lb[01]ELi\d+E", 256), (r"qp_group16_kernelI[df]", 256), (r"qp_list_kernelI[df]", 256),
    # the host sizes fp64 tiles for THREE resident workgroups per CU (wbc_api.cpp: one round of 768): that needs <= 168 registers
    (r"qp_tile_kernelIdLb[01]ELi(32|36|40|44|48|52|56|60|64)ELb0E", 168),
    # ... and since the QP weights stay scalar in the tiled body, the fp32 tiles and the hand-over list kernel hold three wavefronts per SIMD too
    (r"qp_tile_kernelIfLb[01]ELi(32|36|40|44|48|52|56|60|64|72|80|88|96|104|112|120|128)ELb0E", 168), (r"qp_list_kernelI[df]Lb[01]ELb0E", 168),
    # staged tiles run twelve wavefronts per workgroup, one workgroup per CU: three wavefronts per SIMD
    (r"qp_stile_kernelIfLb[01]ELi12ELi[123]E", 168),
    # the tile tick: sweep | observer roles and the QP stage in one workgroup, two wavefronts per SIMD
    (r"tile_tick_kernelIfLi2ELi[234]E", 256), (r"tile_tick_kernelIdLi1ELi[2-7]E", 256),
    # (the WARM list kernel -- a few per cent of a warm-started batch at most -- carries the block set-up: two wavefronts per SIMD)
    (r"qp_general_kernelI[df]", 128),
    (r"rnea_step_kernelIdLi(2|10|34|42)ELi256", 256),
]


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_hot_kernels_keep_their_register_budget(tmp_path):
    import re
    spec = importlib.util.spec_from_file_location("spill_lint", os.path.join(ROOT, "tools", "spill_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.resources(mod.compile_asm(str(tmp_path / "wbc.s")))
    for pat, regs in BUDGET:
        hits = {k: v for k, v in res.items() if re.search(pat, k)}
        assert hits, pat
        for k, v in hits.items():
            # no spill traffic: no private segment at all, or one that no instruction of the kernel touches (hipcc leaves a 36-byte frame object behind in
            # one fused instantiation: SGPR spills it then served from VGPR lanes)
            assert v["scratch"] == 0 or (v["scratch"] <= 64 and v["scratch_insts"] == 0), (k, v)
            assert v["vgpr"] + v["agpr"] <= regs, (k, v)


def test_lint_flags_the_pattern(tmp_path):
    spec = importlib.util.spec_from_file_location("spill_lint", os.path.join(ROOT, "tools", "spill_lint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    p = tmp_path / "k.s"
    p.write_text("_Z1kv:\n\ts_and_saveexec_b64 s[6:7], vcc\n.LBB0_2:\n\ts_andn2_b64 exec, exec, s[8:9]\n\ts_cbranch_execnz .LBB0_2\n"
                 ".LBB0_3:\n\tscratch_store_dwordx2 off, v[22:23], off offset:8 ; 8-byte Folded Spill\n\ts_or_b64 exec, exec, s[6:7]\n"
                 "\tscratch_store_dwordx2 off, v[2:3], off offset:16 ; 8-byte Folded Spill\n\ts_endpgm\n")
    bad = mod.lint(str(p))
    assert len(bad) == 1 and bad[0][1] == ".LBB0_3" and "v[22:23]" in bad[0][3]
