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
