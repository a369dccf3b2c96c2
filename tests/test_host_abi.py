"""CPU: host-side logic of the product library -- URDF reader parity with the independent Python parser,
C-ABI symbol coverage against include/wbc_hip.h, and error behaviour.  No compute calls (no GPU here)."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import wbc_quadruped_dob_amd as W

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_every_declared_symbol_is_exported(hip_lib):
    hdr = open(os.path.join(ROOT, "include", "wbc_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(wbc_[a-z_0-9]+)\s*\(", hdr))
    assert len(declared) >= 18, declared
    nm = subprocess.run(["nm", "-D", "--defined-only", W.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\bT (wbc_[a-z_0-9]+)\b", nm))
    assert declared <= exported, declared - exported
    for name in declared:
        assert getattr(hip_lib, name) is not None


def test_urdf_reader_matches_python_parser(hip_lib, flat_model):
    m = W.Model.from_urdf(W.SYNTHETIC_URDF)
    assert (m.nb, m.nq, m.nv, m.nj, m.nf) == (13, 19, 18, 12, 4)
    f = m.flat()
    for k in ("parent", "foot_body"):
        np.testing.assert_array_equal(f[k], flat_model[k])
    for k in ("Rt", "rt", "axis", "mass", "com", "Ic", "foot_off", "gravity"):
        np.testing.assert_allclose(np.asarray(f[k]).reshape(-1), np.asarray(flat_model[k]).reshape(-1), rtol=0, atol=1e-14, err_msg=k)
    assert f["joint_names"] == flat_model["joint_names"]
    assert f["foot_links"] == flat_model["foot_links"]
    assert abs(m.total_mass - float(flat_model["mass"].sum())) < 1e-12


def test_explicit_foot_links_and_flat_roundtrip(hip_lib, flat_model):
    feet = ["back_right_foot", "front_left_foot", "back_left_foot", "front_right_foot"]
    m = W.Model.from_urdf(W.SYNTHETIC_URDF, foot_links=feet)
    f = m.flat()
    assert list(f["foot_body"]) == [12, 3, 9, 6] and f["foot_links"] == feet
    m2 = W.Model.from_flat(f)
    f2 = m2.flat()
    for k in ("Rt", "rt", "axis", "mass", "com", "Ic", "foot_off"):
        np.testing.assert_array_equal(f2[k], f[k])


def _load(path_or_text, tmp_path, as_text=False):
    if as_text:
        p = tmp_path / "m.urdf"
        p.write_text(path_or_text)
        path_or_text = str(p)
    h = C.c_void_p()
    rc = W.lib().wbc_model_load_urdf(path_or_text.encode(), None, 0, C.byref(h))
    if rc == 0:
        W.lib().wbc_model_free(h)
    return rc, W.lib().wbc_last_error().decode()


def test_urdf_error_codes(hip_lib, tmp_path):
    rc, msg = _load("/nonexistent/dogbot.urdf", tmp_path)
    assert rc == 2 and "cannot open" in msg                       # WBC_E_IO
    rc, msg = _load("<robot name='x'><link name='a'></robot>", tmp_path, True)
    assert rc == 3                                                  # mismatched tag
    rc, msg = _load("<notrobot/>", tmp_path, True)
    assert rc == 3 and "robot" in msg
    rc, msg = _load("<robot><link name='a'/><link name='b'/><joint name='j' type='prismatic'>"
                    "<parent link='a'/><child link='b'/></joint></robot>", tmp_path, True)
    assert rc == 3 and "prismatic" in msg
    rc, msg = _load("<robot><link name='a'/><joint name='j' type='fixed'><parent link='a'/><child link='zz'/></joint></robot>",
                    tmp_path, True)
    assert rc == 3 and "unknown link" in msg
    assert W.lib().wbc_model_load_urdf(None, None, 0, None) == 1    # WBC_E_INVALID


def test_solver_rejects_non_quadruped_and_bad_params(hip_lib, tmp_path):
    two_link = ("<robot><link name='a'><inertial><mass value='1'/><inertia ixx='1' iyy='1' izz='1'/></inertial></link>"
                "<link name='b'><inertial><mass value='1'/><inertia ixx='1' iyy='1' izz='1'/></inertial></link>"
                "<joint name='j' type='revolute'><parent link='a'/><child link='b'/><axis xyz='0 0 1'/></joint></robot>")
    p = tmp_path / "two.urdf"
    p.write_text(two_link)
    m = W.Model.from_urdf(str(p))
    assert (m.nb, m.nf) == (2, 1)
    prm = W.Params.default()
    h = C.c_void_p()
    rc = W.lib().wbc_solver_create(m._h, C.byref(prm), 0, 0, 16, C.byref(h))
    assert rc == 4, W.lib().wbc_last_error()                       # WBC_E_TOPOLOGY before any device is touched
    good = W.Model.from_urdf(W.SYNTHETIC_URDF)
    prm.alpha = 0.0
    assert W.lib().wbc_solver_create(good._h, C.byref(prm), 0, 0, 16, C.byref(h)) == 1
    prm = W.Params.default()
    prm.observer_order = 3
    assert W.lib().wbc_solver_create(good._h, C.byref(prm), 0, 0, 16, C.byref(h)) == 1
    assert W.lib().wbc_solver_create(good._h, C.byref(W.Params.default()), 7, 0, 16, C.byref(h)) == 1  # bad dtype


def test_no_cpu_fallback_without_a_gpu(hip_lib):
    """On a box without a GPU the product must fail loudly, never compute on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    good = W.Model.from_urdf(W.SYNTHETIC_URDF)
    h = C.c_void_p()
    rc = W.lib().wbc_solver_create(good._h, C.byref(W.Params.default()), 0, 0, 16, C.byref(h))
    assert rc == 5 and "no CPU fallback" in W.lib().wbc_last_error().decode()
    with pytest.raises(RuntimeError):
        W.Solver(good)


def test_product_does_not_reference_the_oracle():
    """No product source may include, import or link anything under oracle/."""
    pkg = os.path.join(ROOT, "wbc_quadruped_dob_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h")) or fn == "Makefile":
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "wbc_oracle" not in txt and "from oracle" not in txt and "import oracle" not in txt, fn
    ldd = subprocess.run(["ldd", W.LIB_PATH], capture_output=True, text=True).stdout
    assert "oracle" not in ldd


def test_params_default_values(hip_lib):
    p64, p32 = W.Params.default("f64"), W.Params.default("f32")
    assert p64.alpha == 1e-3 and p64.qp_tol == 1e-9 and p32.qp_tol == 1e-3 and p64.observer_order == 0
    assert list(p64.S) == [1.0] * 6 and p64.K1[17] == 50.0 and p64.K2[0] == 200.0


def test_solver_options_struct_and_validation(hip_lib):
    """wbc_solver_options carries every kernel-selection switch (the library reads no environment variable)."""
    o = W.SolverOptions.default()
    assert o.struct_size == C.sizeof(W.SolverOptions)
    assert (o.fused_max, o.rollout_persistent, o.rollout_spw, o.obs_split_min, o.one_zerocopy, o.timing_mode, o.qp_tile, o.obs_split_serial, o.qp_lane, o.f32_pack2, o.keep_structural) == (-1, 1, 0, -1, 3, 0, 0, 1, 0, 0, 0)
    good = W.Model.from_urdf(W.SYNTHETIC_URDF)
    h = C.c_void_p()
    prm = W.Params.default()
    bad = W.SolverOptions.default()
    bad.struct_size = 0
    assert W.lib().wbc_solver_create_ex(good._h, C.byref(prm), 0, 0, 16, C.byref(bad), C.byref(h)) == 1
    bad = W.SolverOptions.default()
    bad.rollout_spw = 5
    assert W.lib().wbc_solver_create_ex(good._h, C.byref(prm), 0, 0, 16, C.byref(bad), C.byref(h)) == 1
    assert "rollout_spw" in W.lib().wbc_last_error().decode()
    with pytest.raises(KeyError):
        W.SolverOptions.make({"no_such_switch": 1})
    src = open(os.path.join(ROOT, "wbc_quadruped_dob_amd", "csrc", "wbc_api.cpp")).read() + \
        open(os.path.join(ROOT, "wbc_quadruped_dob_amd", "csrc", "wbc_multi.cpp")).read()
    assert "getenv" not in src


def test_shard_range_c_abi_matches_python_sharding(hip_lib):
    from wbc_quadruped_dob_amd.sharding import shard_range
    for n_total in (0, 1, 7, 1001, 262144):
        for world in (1, 2, 3, 8):
            spans = [W.shard_range(n_total, world, r) for r in range(world)]
            assert spans == [shard_range(n_total, world, r) for r in range(world)]
            assert sum(c for _, c in spans) == n_total and spans[0][1] == max(c for _, c in spans)
    st, cnt = C.c_size_t(), C.c_size_t()
    assert W.lib().wbc_shard_range(10, 0, 0, C.byref(st), C.byref(cnt)) == 1
    assert W.lib().wbc_shard_range(10, 2, 2, C.byref(st), C.byref(cnt)) == 1


def test_multi_create_argument_errors(hip_lib):
    good = W.Model.from_urdf(W.SYNTHETIC_URDF)
    prm = W.Params.default()
    h = C.c_void_p()
    dup = (C.c_int * 2)(0, 0)
    rc = W.lib().wbc_multi_create(good._h, C.byref(prm), 0, dup, 2, 64, W.GATHER_RCCL, None, C.byref(h))
    assert rc == 1 and "distinct devices" in W.lib().wbc_last_error().decode()
    assert W.lib().wbc_multi_create(good._h, C.byref(prm), 0, dup, 0, 64, 0, None, C.byref(h)) == 1
    assert W.lib().wbc_multi_create(good._h, C.byref(prm), 0, dup, 2, 64, 9, None, C.byref(h)) == 1
    assert W.lib().wbc_multi_size(None) == 0 and W.lib().wbc_multi_rccl_ranks(None) == 0
    # the tick entry points refuse null handles / null argument arrays before they touch a device
    L = W.lib()
    assert L.wbc_multi_step_batch(None, 8, None, None, None) == 1
    assert L.wbc_multi_step_batch_warm(None, 8, None, None, None, None) == 1
    L.wbc_step_batch_warm.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    assert L.wbc_step_batch_warm(None, 8, None, None, None, None, None, None) == 1
    plan = W.TickPlan()
    plan.struct_size = C.sizeof(W.TickPlan)
    assert L.wbc_plan_tick(7, 0, None, 4096, 1, 1, 0, C.byref(plan)) == 1            # no such scalar type
    assert L.wbc_plan_tick(W.F64, 3, None, 4096, 1, 1, 0, C.byref(plan)) == 1        # no such observer order
    assert L.wbc_plan_tick(W.F64, 1, None, 4096, 1, 1, 1, C.byref(plan)) == 0 and plan.fused == 1 and plan.qp_warm == 1
    # an older caller's smaller wbc_tick_plan: only its bytes are written
    plan2 = W.TickPlan()
    plan2.struct_size = C.sizeof(W.TickPlan) - C.sizeof(C.c_int)
    plan2.qp_warm = 77
    assert L.wbc_plan_tick(W.F64, 1, None, 4096, 1, 1, 1, C.byref(plan2)) == 0 and plan2.fused == 1 and plan2.qp_warm == 77


def test_ros_section_compiles_against_stubs(hip_lib, tmp_path):
    """The WBC_WITH_ROS adaptors of include/wbc/quadruped_wbc.hpp against field-layout STUBS of the message headers
    (tests/stubs/: labelled stubs, they pin nothing about the reference's topics).  Catches rot of code that no ROS-less
    build would otherwise ever compile."""
    exe = str(tmp_path / "ros_check")
    libdir = os.path.dirname(W.LIB_PATH)
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "tests", "stubs"),
                    os.path.join(ROOT, "tests", "stubs", "ros_section_check.cpp"), "-o", exe, "-L" + libdir, "-lwbc_hip",
                    "-Wl,-rpath," + libdir], check=True, capture_output=True, text=True)
    run = subprocess.run([exe], capture_output=True, text=True)
    assert run.returncode == 0 and "ros adaptors ok" in run.stdout, run.stdout + run.stderr


def test_bench_refuses_more_gpus_than_present():
    """`python bench.py --gpus N` started bare launches its own ranks; with fewer GPUs than asked for it must exit
    non-zero with a clear message instead of asserting or hanging."""
    import torch
    n = torch.cuda.device_count() + 1
    if n < 2:
        n = 2
    run = subprocess.run(["python", os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env={k: v for k, v in os.environ.items() if k != "WORLD_SIZE"})
    assert run.returncode == 3 and "GPU(s) visible" in run.stderr


def test_abi_version_is_the_same_everywhere(hip_lib):
    """The number the library returns, the one the header documents and the one __graft_entry__.build() asserts (a bump that
    forgot the last one would fail the driver's build check, not a test)."""
    import re
    hdr = open(os.path.join(ROOT, "include", "wbc_hip.h")).read()
    doc = int(re.search(r"int wbc_abi_version\(void\); /\* (\d+) \*/", hdr).group(1))
    entry = int(re.search(r"wbc_abi_version\(\) == (\d+)", open(os.path.join(ROOT, "__graft_entry__.py")).read()).group(1))
    assert hip_lib.wbc_abi_version() == doc == entry


def test_dense_qp_argument_checks_need_no_gpu(hip_lib):
    """wbc_qp_dense_batch validates sizes and pointers before any HIP call (no device in this container): bad dtype, sizes outside
    1 <= n <= 36, 0 <= meq <= m <= 64, null pointers, negative iteration limit -> WBC_E_INVALID; an empty batch is a no-op."""
    import ctypes as C
    f = hip_lib.wbc_qp_dense_batch
    f.argtypes = [C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 4 + [C.c_int, C.c_double] + [C.c_void_p] * 5
    p = C.c_void_p(16)   # any non-null address: never dereferenced on these paths
    ok_args = lambda **kw: [kw.get("dtype", 0), kw.get("N", 4), kw.get("n", 12), kw.get("m", 24), kw.get("meq", 0), kw.get("H", p), p, kw.get("C", p), p,
                            kw.get("max_iter", 100), kw.get("tol", 1e-9), p, None, kw.get("status", p), None, None]
    INVALID = 1
    for bad in (dict(dtype=2), dict(n=0), dict(n=37), dict(m=65), dict(m=-1), dict(meq=25), dict(meq=-1), dict(H=None), dict(C=None),
                dict(status=None), dict(max_iter=-1), dict(tol=-1.0)):
        assert f(*ok_args(**bad)) == INVALID, bad
        assert hip_lib.wbc_last_error()
    assert f(*ok_args(N=0)) == 0
    assert f(*ok_args(N=0, m=0, C=None)) == 0      # no rows: C and d may be null


def test_dispatch_thresholds_cover_the_documented_switches(hip_lib):
    """The defaults as the planner reports them; a deliberate change of a default changes this list.  Round 5: mid-size observer-on
    ticks with matrix outputs run the observer update and the sweep as the two roles of one launch (front = 4): fp64 12 289 .. 14 336, fp32 .. 32 768."""
    import wbc_quadruped_dob_amd as W
    assert W.dispatch_thresholds("f64", 0) == [4225, 8193, 28673, 65536, 106496]      # (round 6: 8 193 .. 28 672 states as one round of tile-tick workgroups, in front of the one-launch tick; 4 225 .. 8 192: the one-launch tick as 32-state workgroups)
    assert W.dispatch_thresholds("f64", 0, options={"tile_tick": -1}) == [11265, 14336, 65536, 106496]
    assert [(W.plan_tick(n, "f64", 0)["fused"], W.plan_tick(n, "f64", 0)["qp_tile"]) for n in (4096, 8191, 8192, 8193, 12288, 12289, 28672, 28673)] == [(1, 0), (3, 0), (3, 0), (2, 48), (2, 48), (2, 64), (2, 112), (0, 40)]
    # round 6: fp64 observer-on batches of 8 193 .. 196 608 states run the tile tick too (32 / 48 / 64-state workgroups; 64-state ones in rounds beyond 16 384 states)
    assert W.dispatch_thresholds("f64", 1) == [8193, 196609]
    assert W.dispatch_thresholds("f64", 1, options={"tile_tick": -1}) == [12289, 14336, 14337, 20480, 65536, 106496]
    assert [(W.plan_tick(n, "f64", 1)["fused"], W.plan_tick(n, "f64", 1)["qp_tile"]) for n in (8192, 8193, 12288, 12289, 16384, 65536, 196608, 196609)] == [
        (1, 0), (2, 48), (2, 48), (2, 64), (2, 64), (2, 64), (2, 64), (0, 0)]
    # round 6: even fp32 observer-on batches from 8 194 states on run the tile tick (one launch of 64 / 96 / 128-state workgroups; up to 12 288 states in front of the one-launch tick); staged QP tiles up to 49 152
    assert W.dispatch_thresholds("f32", 1) == [8194]
    assert W.dispatch_thresholds("f32", 1, options={"tile_tick": -1}) == [12289, 16384, 32769, 33792, 49153, 65537, 131072, 212992]
    assert [(W.plan_tick(n, "f32", 1)["fused"], W.plan_tick(n, "f32", 1)["qp_tile"]) for n in (8192, 8194, 12288, 12290, 16384, 16386, 24576, 24578, 32768, 32770, 32771, 262144)] == [
        (1, 0), (2, 64), (2, 64), (2, 64), (2, 64), (2, 96), (2, 96), (2, 128), (2, 128), (2, 64), (0, 132), (2, 64)]      # (beyond one round of workgroups: 64-state ones, two per CU)
    assert W.dispatch_thresholds("f32", 0) == [4225, 16385, 32768, 49153, 65537, 131072, 212992]      # (round 6: 4 225 .. 16 384 states as 32-state workgroups of the one-launch tick)
    assert W.dispatch_thresholds("f32", 0, options={"fused_pair": -1}) == [11265, 16384, 32768, 49153, 65537, 131072, 212992]
    assert [(W.plan_tick(n, "f32", 0)["qp_body"], W.plan_tick(n, "f32", 0)["qp_tile"]) for n in (16382, 16386, 32768, 49152, 49154)] == [(0, 0), (2, 68), (2, 128), (2, 192), (0, 72)]
    assert W.plan_tick(32768, "f32", 1, options={"tile_tick": -1}) == dict(fused=0, front=4, qp=1, qp_tile=128, qp_body=2, sweep_pack2=1, sweep_block=64, qp_warm=0)
    assert [W.plan_tick(n, "f64", 1, options={"tile_tick": -1})["front"] for n in (12288, 12289, 14336, 14337, 20480)] == [0, 4, 4, 0, 2]
    assert [W.plan_tick(n, "f32", 1, options={"tile_tick": -1})["front"] for n in (12290, 12291, 32768, 32770, 33792)] == [4, 0, 4, 0, 2]
    assert W.plan_tick(20000, "f32", 1)["sweep_pack2"] == 1 and W.plan_tick(20000, "f32", 1, options={"obs_colaunch": -1, "tile_tick": -1})["sweep_pack2"] == 0
    assert W.plan_tick(262144, "f64", 0)["qp"] == 2 and W.plan_tick(262144, "f32", 1, options={"tile_tick": -1}) == dict(
        fused=0, front=2, qp=2, qp_tile=0, qp_body=0, sweep_pack2=1, sweep_block=256, qp_warm=0)
    # options move the switches, and the list follows
    assert W.dispatch_thresholds("f64", 0, options={"qp_lane": -1, "qp_tile": -1, "fused_max": 0}) == [65536]
    # ticks whose caller passes no M / h / Jc buffers: rnea_step front half; observer kernel + observer-free rnea_step from 16 384 fp64 / 32 768 fp32 states
    assert W.dispatch_thresholds("f64", 1, want_mats=False) == [14336, 16384, 106496]
    assert W.dispatch_thresholds("f32", 1, want_mats=False) == [16384, 32768, 49153, 65537, 212992]
    assert [W.plan_tick(n, "f64", 1, want_mats=False)["front"] for n in (9000, 16383, 16384, 262144)] == [1, 1, 3, 3]
    assert W.plan_tick(262144, "f64", 0, want_mats=False)["front"] == 1
    # warm-started ticks (wbc_step_batch_warm): fused, warm one-wavefront kernel, cold tiles that only report the sets, warm per-lane pair
    # (round 6: behind the one-launch warm tick and below the warm per-lane pair the warm call runs the cold, set-reporting tile tick)
    assert W.dispatch_thresholds("f64", 1, warm=True) == [12289, 53248, 65536]
    assert W.dispatch_thresholds("f64", 1, warm=True, options={"tile_tick": -1}) == [12289, 14337, 20480, 24576, 53248, 65536]
    assert W.dispatch_thresholds("f32", 1, warm=True) == [12289, 81920, 131072]
    assert W.dispatch_thresholds("f32", 1, warm=True, options={"tile_tick": -1}) == [12289, 30720, 32769, 33792, 36864, 131072]
    assert [(W.plan_tick(n, "f64", 1, warm=True)["fused"], W.plan_tick(n, "f64", 1, warm=True)["qp_warm"]) for n in (4096, 12288, 13000, 30000, 53247, 60000)] == [(1, 1), (1, 1), (2, 0), (2, 0), (2, 0), (0, 1)]
    assert [W.plan_tick(n, "f64", 1, warm=True, options={"tile_tick": -1})["qp_warm"] for n in (4096, 13000, 20000, 30000, 60000)] == [1, 1, 1, 0, 1]
    assert [W.plan_tick(n, "f64", 1, warm=True, options={"tile_tick": -1})["qp"] for n in (13000, 20000, 30000, 60000)] == [0, 0, 1, 2]
    assert [W.plan_tick(n, "f64", ob)["fused"] for n, ob in ((8192, 0), (8193, 0), (11264, 0), (8192, 1), (12288, 1), (12289, 1))] == [3, 2, 2, 1, 2, 2]
    assert [W.plan_tick(n, "f64", 0, options={"fused_max": 11264})["fused"] for n in (8193, 11264, 11265)] == [1, 1, 2]   # (a caller who names the one-launch tick's limit keeps it)




def test_design_md_carries_the_current_dispatch_table(hip_lib):
    """DESIGN.md section 5 is GENERATED from the planner (tools/gen_dispatch_table.py, refreshed by tools/update_design.py): a moved threshold that is
    not carried into the document fails here.  Also: the document stays a design record -- at most 50 kB."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("update_design", os.path.join(ROOT, "tools", "update_design.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    text = open(os.path.join(ROOT, "DESIGN.md")).read()
    m = re.search(r"<!-- BEGIN generated by tools/gen_dispatch_table.py -->.*?<!-- END generated -->", text, flags=re.S)
    assert m, "DESIGN.md has lost its generated dispatch table"
    assert m.group(0) == mod.generated(), "DESIGN.md section 5 is stale: run python tools/update_design.py"
    assert len(text.encode()) <= 50 * 1024


@pytest.mark.parametrize("threads,tickets,spin_us,pause_us", [(8, 20000, 200, 0), (8, 400, 0, 0), (3, 300, 0, 150), (2, 300, 50, 120), (16, 2000, 20, 0)])
def test_issue_threads_run_every_ticket_once(hip_lib, threads, tickets, spin_us, pause_us):
    """wbc_multi_*'s per-shard issue threads (csrc/wbc_multi.cpp, IssuePool), without a device: every thread runs every ticket exactly once and in order, an
    error of one thread reaches the caller with its message -- spinning (a tick loop), always parked (spin_us = 0) and parking between tickets (pause_us)."""
    hip_lib.wbc_multi_selftest_issue.argtypes = [C.c_int] * 4
    rc = hip_lib.wbc_multi_selftest_issue(threads, tickets, spin_us, pause_us)
    assert rc == 0, hip_lib.wbc_last_error()
