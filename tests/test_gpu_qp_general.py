"""GPU: the general dense QP kernel (wbc_qp_dense_batch -> csrc/qp_general.hip.hpp: run-time sizes n <= 36, m <= 64, equality rows,
one QP per wavefront with its factors in LDS) against the oracle's general solver (oracle/qp_general.hpp), through the C-ABI.

* random strictly convex problems of every size class, fp64: status and iteration counts EQUAL the oracle's (same method, same
  tests), x within 1e-9, multipliers within 1e-7; fp32 against the fp64 oracle at a stated fp32 tolerance;
* the controller's own GRF QPs (assembled per state by the oracle from the kernel-computed foot positions) through the general
  kernel reproduce the forces of the structured kernels inside wbc_step_batch;
* status codes (infeasible, iteration limit, not positive definite), ragged batch sizes, equality-only and constraint-free problems;
* an oracle-free KKT check at the largest size on 2 x 10^4 problems."""
import numpy as np
import pytest

from tests.util import random_problem, relerr, to_dev, to_host
from wbc_quadruped_dob_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_cuda():
    import torch
    assert torch.cuda.is_available(), "GPU test run without a GPU"
    return torch


def _batch(rng, N, n, m, meq, cond=1e3):
    probs = [random_problem(rng, n, m, meq, cond) for _ in range(N)]
    return tuple(np.stack([p[k] for p in probs]) for k in range(4))


def _gpu(torch, H, g, C, d, meq, dtype=None, **kw):
    import wbc_quadruped_dob_amd as W
    td = torch.float64 if dtype in (None, "f64") else torch.float32
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(td).cuda().contiguous()
    m = d.shape[1]
    out = W.qp_dense_batch(dev(H), dev(g), dev(C) if m else None, dev(d) if m else None, meq=meq, **kw)
    torch.cuda.synchronize()
    return {k: (None if v is None else v.cpu().numpy()) for k, v in out.items()}


@pytest.mark.parametrize("n,m,meq,N", [(1, 2, 0, 7), (3, 5, 1, 130), (12, 24, 0, 1001), (12, 24, 4, 300), (20, 30, 6, 257), (36, 48, 0, 301),
                                       (36, 48, 10, 129), (30, 48, 30, 64), (9, 0, 0, 50), (7, 7, 7, 33),
                                       (30, 58, 18, 200), (36, 64, 12, 130)])   # (the size of a whole-body QP over accelerations and forces; the largest)
def test_random_problems_vs_oracle_fp64(torch_cuda, n, m, meq, N):
    from oracle import oracle_py as O
    rng = np.random.default_rng(100 * n + m + meq)
    H, g, C, d = _batch(rng, N, n, m, meq)
    out = _gpu(torch_cuda, H, g, C, d, meq, max_iter=400, tol=1e-10)
    xr, lr, sr, ir = O.qp_general(H, g, C, d, meq, max_iter=400, tol=1e-10)
    assert np.array_equal(out["status"], sr) and (sr == 0).all()
    assert np.mean(out["iters"] != ir) < 0.01        # (a tie in a ratio test can resolve the other way round; never seen on these seeds)
    assert relerr(out["x"], xr) < 1e-9
    if m:
        assert np.abs(out["lam"] - lr).max() < 1e-7 * (1 + np.abs(lr).max())
        assert ir.max() > 0


def test_fp32_vs_fp64_oracle(torch_cuda):
    from oracle import oracle_py as O
    rng = np.random.default_rng(11)
    H, g, C, d = _batch(rng, 400, 12, 24, 2, cond=50.0)
    out = _gpu(torch_cuda, H, g, C, d, 2, dtype="f32", max_iter=200, tol=1e-4)
    xr, _, sr, _ = O.qp_general(H, g, C, d, 2, max_iter=200, tol=1e-10)
    ok = (out["status"] == 0) & (sr == 0)
    assert ok.mean() > 0.99
    assert relerr(out["x"][ok], xr[ok]) < 2e-3       # fp32 arithmetic throughout (cond(H) = 50): the stated fp32 tolerance of this path
    s = np.einsum("kij,kj->ki", C, out["x"].astype(np.float64)) - d
    assert np.abs(s[ok, :2]).max() < 1e-2 and s[ok, 2:].min() > -1e-2


def test_grf_qps_through_the_general_kernel_match_the_structured_path(torch_cuda, gpu_model, oracle):
    """The controller's QP (a7) assembled per state by the oracle from the HIP path's own foot positions, solved by the general
    kernel, against f of wbc_step_batch (structured wrench-space kernels) on the same states: one solution, two very different
    kernels."""
    import wbc_quadruped_dob_amd as W
    from oracle import oracle_py as O
    torch = torch_cuda
    n = 600
    B = synth.make_batch(3, n, gpu_model.total_mass, rank=91)
    B["w_des"][: n // 2, 0:2] += np.random.default_rng(2).uniform(-80, 80, (n // 2, 2))
    P = synth.default_params(observer_order=0)
    solver = W.Solver(gpu_model, W.Params.from_dict(P, "f64"), dtype="f64", device=0, max_batch=n)
    td = torch.float64
    dv = lambda k: to_dev(B[k], torch, td)
    out = solver.step(dv("q"), dv("v"), dv("w_des"), dv("vdot_des"), dv("normals"), dv("mu"), torch.from_numpy(B["mask"]).cuda(), want_mats=True)
    torch.cuda.synchronize()
    f = to_host(out["f"])
    pf = to_host(out["pf"])
    st = out["status"].cpu().numpy()
    seen = 0
    for mask in np.unique(B["mask"]):
        idx = np.nonzero((B["mask"] == mask) & (st == 0))[0]
        if len(idx) == 0 or mask == 0:
            continue
        feet = [k for k in range(4) if (int(mask) >> k) & 1]
        qs = [O.qp_assemble(P, 4, int(mask), B["q"][s, :3], pf[s], B["normals"][s], B["mu"][s], B["w_des"][s]) for s in idx]
        H, g, C, d = (np.stack([q[k] for q in qs]) for k in range(4))
        res = _gpu(torch, H, g, C, d, 0, max_iter=P["max_iter"], tol=P["qp_tol"])
        assert (res["status"] == 0).all()
        fs = np.zeros((len(idx), 12))
        for j, k in enumerate(feet):
            fs[:, 3 * k:3 * k + 3] = res["x"][:, 3 * j:3 * j + 3]
        assert relerr(fs, f[idx]) < 1e-9, mask
        seen += len(idx)
    assert seen > 0.9 * n


def test_status_codes_ragged_sizes_and_edges(torch_cuda):
    from oracle import oracle_py as O
    rng = np.random.default_rng(4)
    N = 67
    H, g, C, d = _batch(rng, N, 8, 12, 0)
    C[3, 0] = -C[3, 1]; d[3, 0] = 1.0; d[3, 1] = 1.0            # row 1 = -row 0, both >= 1: infeasible
    H[5] = np.diag([1.0, 1, 1, -1, 1, 1, 1, 1])                  # not positive definite
    out = _gpu(torch_cuda, H, g, C, d, 0)
    _, _, sr, _ = O.qp_general(H, g, C, d, 0)
    assert np.array_equal(out["status"], sr) and sr[3] == 2 and sr[5] == 3 and (np.delete(sr, [3, 5]) == 0).all()
    assert (out["x"][5] == 0).all()
    lim = _gpu(torch_cuda, H, g, C, d, 0, max_iter=1)
    _, _, sl, il = O.qp_general(H, g, C, d, 0, max_iter=1)
    assert np.array_equal(lim["status"], sl) and np.array_equal(lim["iters"], il) and (sl == 1).any()
    # a dependent equality row that holds is skipped, one that cannot hold makes the problem infeasible
    He, ge = np.tile(np.eye(3), (2, 1, 1)), np.zeros((2, 3))
    Ce = np.tile(np.array([[1.0, 1, 0], [2.0, 2, 0]]), (2, 1, 1))
    de = np.array([[1.0, 2.0], [1.0, 3.0]])
    oe = _gpu(torch_cuda, He, ge, Ce, de, 2)
    assert list(oe["status"]) == [0, 2] and np.allclose(oe["x"][0], [0.5, 0.5, 0]) and oe["lam"][0, 1] == 0
    # no rows at all, no multipliers wanted, a single problem
    one = _gpu(torch_cuda, H[:1], g[:1], np.zeros((1, 0, 8)), np.zeros((1, 0)), 0, want_lambda=False)
    assert one["lam"] is None and one["status"][0] == 0 and one["iters"][0] == 0
    assert np.allclose(np.einsum("ij,j->i", H[0], one["x"][0]), -g[0])


def test_bad_arguments_are_refused(torch_cuda):
    import wbc_quadruped_dob_amd as W
    torch = torch_cuda
    H = torch.eye(37, dtype=torch.float64, device="cuda").repeat(2, 1, 1)
    with pytest.raises(W.WbcError):
        W.qp_dense_batch(H, torch.zeros(2, 37, dtype=torch.float64, device="cuda"))
    H = torch.eye(4, dtype=torch.float64, device="cuda").repeat(2, 1, 1).contiguous()
    g = torch.zeros(2, 4, dtype=torch.float64, device="cuda")
    with pytest.raises(W.WbcError):
        W.qp_dense_batch(H, g, torch.zeros(2, 3, 4, dtype=torch.float64, device="cuda"), torch.zeros(2, 3, dtype=torch.float64, device="cuda"), meq=4)
    with pytest.raises(ValueError):
        W.qp_dense_batch(H, g.float())


def test_kkt_at_the_largest_size_without_the_oracle(torch_cuda):
    """2 x 10^4 problems of 36 variables, 48 rows (8 of them equalities), generated on the GPU: stationarity, feasibility, sign and
    complementarity of the kernel's (x, lambda), computed in torch."""
    import wbc_quadruped_dob_amd as W
    torch = torch_cuda
    N, n, m, meq = 20000, 36, 48, 8
    gen = torch.Generator(device="cuda").manual_seed(5)
    A = torch.randn(N, n, n, dtype=torch.float64, device="cuda", generator=gen)
    H = (A @ A.transpose(1, 2)) / n + torch.eye(n, dtype=torch.float64, device="cuda")
    H = (0.5 * (H + H.transpose(1, 2))).contiguous()
    xf = torch.randn(N, n, dtype=torch.float64, device="cuda", generator=gen)
    C = torch.randn(N, m, n, dtype=torch.float64, device="cuda", generator=gen).contiguous()
    slack = torch.rand(N, m, dtype=torch.float64, device="cuda", generator=gen)
    slack[:, :meq] = 0
    d = (torch.einsum("kij,kj->ki", C, xf) - slack).contiguous()
    g = (-torch.einsum("kij,kj->ki", H, xf + 2 * torch.randn(N, n, dtype=torch.float64, device="cuda", generator=gen))).contiguous()
    out = W.qp_dense_batch(H, g, C, d, meq=meq, max_iter=500, tol=1e-10)
    torch.cuda.synchronize()
    assert (out["status"] == 0).all()
    x, lam = out["x"], out["lam"]
    s = torch.einsum("kij,kj->ki", C, x) - d
    sc = 1 + g.abs().amax(1)
    stat = (torch.einsum("kij,kj->ki", H, x) + g - torch.einsum("kij,ki->kj", C, lam)).abs().amax(1) / sc
    assert stat.max().item() < 1e-9
    assert s[:, :meq].abs().max().item() < 1e-8 and s[:, meq:].min().item() > -1e-8
    assert lam[:, meq:].min().item() >= 0
    assert ((lam[:, meq:] * s[:, meq:]).abs().amax(1) / sc).max().item() < 1e-8
    assert out["iters"].double().mean().item() > 10
