"""GPU parity tests (through the C-ABI) of the volume kernels against the CPU oracle,
plus size-independent properties at BASELINE.json's full config-2 size.

fp64 tolerance: the fused kernel re-associates the reference's 27 separately rounded
passes, so parity is stated relative to ||Au||_inf: 1e-12 (the reference's own
cross-implementation precedent is 1e-13 absolute per node, d4est_test_laplacian_speedup.c:485)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-12


def _t(a, dev):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


def _plan(m, J, rst):
    from disco4est_amd import Plan
    p = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, m.quad_type)
    p.set_geometry(J, rst)
    return p


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


CASES = [
    # (level, deg, inc, quad_type, curved)
    (1, 1, 0, 0, True), (1, 2, 0, 0, True), (1, 3, 0, 0, False), (1, 3, 0, 0, True), (1, 4, 0, 0, True),
    (1, 5, 0, 0, True), (1, 6, 0, 0, True), (1, 7, 0, 0, False), (1, 7, 0, 0, True), (1, 8, 0, 0, True),
    (1, 9, 0, 0, True), (1, 11, 0, 0, True), (0, 15, 0, 0, True), (1, 12, 0, 0, True),
    (1, 2, 3, 0, True), (1, 3, 1, 0, True), (1, 7, 1, 0, True), (1, 7, 2, 0, True), (1, 3, 2, 0, True),
    (1, 3, 0, 1, True), (1, 7, 0, 1, True), (1, 2, 1, 1, True),
    (0, 17, 0, 0, True), (0, 19, 0, 0, False), (1, 5, 2, 0, True), (1, 1, 1, 0, True), (0, 16, 0, 0, True), (0, 18, 0, 0, True),
    (1, 13, 0, 0, True), (0, 19, 1, 0, True),
]


@pytest.mark.parametrize("level,deg,inc,qt,curved", CASES)
def test_stiffness_parity(gpu, hiplib, oracle, level, deg, inc, qt, curved):
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg, deg_quad_inc=inc, quad_type=qt)
    mp = M.SineMap(0.06) if curved else None
    J, rst = m.geometry(mp)
    u = m.field(mp)
    ref = oracle.apply_stiffness(m, J, rst, u, nthreads=8)
    plan = _plan(m, J, rst)
    du = _t(u, gpu)
    dAu = torch.full_like(du, float("nan"))  # the apply must overwrite every entry
    plan.apply_stiffness_matrix(du, dAu)
    torch.cuda.synchronize()
    got = dAu.cpu().numpy()
    assert np.isfinite(got).all()
    assert _rel(got, ref) <= RTOL
    # host-pointer convenience path gives the same bits
    np.testing.assert_array_equal(plan.apply_stiffness_matrix_host(u), got)
    plan.destroy()


@pytest.mark.parametrize("deg,inc", [(1, 0), (3, 0), (5, 0), (7, 0), (3, 2), (5, 2), (2, 0), (6, 1)])
@pytest.mark.parametrize("tune", [(0, 0), (1, 0), (0, 1), (1, 1), (0, 2), (0, 3), (0, 11)])
def test_stiffness_kernel_variants(gpu, hiplib, oracle, deg, inc, tune):
    """every tuning variant (3-buffer / prefetch / single-wave / two-wave kernels) gives the oracle's answer"""
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(1, deg, deg_quad_inc=inc, count=7)
    mp = M.SineMap(0.06)
    J, rst = m.geometry(mp)
    u = m.field(mp)
    ref = oracle.apply_stiffness(m, J, rst, u, nthreads=4)
    plan = _plan(m, J, rst)
    plan.set_tuning(0, tune[0]); plan.set_tuning(1, tune[1])
    du = _t(u, gpu); dAu = torch.full_like(du, float("nan"))
    plan.apply_stiffness_matrix(du, dAu)
    assert _rel(dAu.cpu().numpy(), ref) <= RTOL


@pytest.mark.parametrize("deg,inc", [(8, 0), (9, 0), (11, 0), (12, 0), (13, 0), (14, 0), (15, 0), (8, 1)])
@pytest.mark.parametrize("bigp", [0, 1, 2])
def test_stiffness_high_p_variants(gpu, hiplib, oracle, deg, inc, bigp):
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(1, deg, deg_quad_inc=inc, count=3)
    mp = M.SineMap(0.06)
    J, rst = m.geometry(mp)
    u = m.field(mp)
    ref = oracle.apply_stiffness(m, J, rst, u, nthreads=4)
    plan = _plan(m, J, rst)
    plan.set_tuning(4, bigp)
    du = _t(u, gpu); dAu = torch.full_like(du, float("nan"))
    plan.apply_stiffness_matrix(du, dAu)
    assert _rel(dAu.cpu().numpy(), ref) <= RTOL


def test_stiffness_p15_matrix_core_kernel_persistent(gpu, hiplib, oracle):
    """p = 15 (config 5's degree): the v_mfma_f64_16x16x4 kernel (default at deg = deg_quad = 15) on more elements than CUs -- every
    workgroup loops over several elements with the next element's slabs in flight -- curved metric, against the oracle."""
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(3, 15, count=300)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp)
    u = m.field(mp)
    ref = oracle.apply_stiffness(m, J, rst, u, nthreads=8)
    plan = _plan(m, J, rst)
    du = _t(u, gpu); dAu = torch.full_like(du, float("nan"))
    plan.apply_stiffness_matrix(du, dAu)
    assert "mfma16" in plan.last_kernel()
    got = dAu.cpu().numpy()
    per_elem = np.abs(got - ref).reshape(300, -1).max(axis=1) / np.abs(ref).reshape(300, -1).max(axis=1)
    assert per_elem.max() <= RTOL, (int(per_elem.argmax()), per_elem.max())
    plan.set_tuning(4, 1)           # the vector-ALU kernel stays selectable and agrees
    dAu2 = torch.full_like(du, float("nan"))
    plan.apply_stiffness_matrix(du, dAu2)
    assert "wave_kernel" in plan.last_kernel()
    assert _rel(dAu2.cpu().numpy(), got) <= RTOL
    plan.destroy()


def test_stiffness_mixed_p_parity(gpu, hiplib, oracle):
    """config-4 style: mixed p = 3..9 in one plan (degree-bucketed launches)."""
    import torch
    from disco4est_amd import mesh as M
    deg = 3 + (np.arange(64) * 5) % 7
    m = M.BrickMesh(2, deg)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp)
    u = m.field(mp)
    ref = oracle.apply_stiffness(m, J, rst, u, nthreads=8)
    plan = _plan(m, J, rst)
    du = _t(u, gpu); dAu = torch.full_like(du, float("nan"))
    plan.apply_stiffness_matrix(du, dAu)
    got = dAu.cpu().numpy()
    assert _rel(got, ref) <= RTOL
    # per-element check so a small-p element cannot hide behind a large-p norm
    for e in range(m.n_elements):
        s = m.nodal_stride[e]; n3 = (deg[e] + 1) ** 3
        assert _rel(got[s:s + n3], ref[s:s + n3]) <= 10 * RTOL


def test_golden_probe_on_gpu(gpu, hiplib):
    """The survey-time outputs of the real reference, reproduced by the HIP kernel."""
    import json, os, torch
    from disco4est_amd import Plan, table
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "survey_probe.json")))
    h = g["h"]
    for case in g["cases"]:
        p = case["p"]; n = p + 1
        x = table("lobatto_nodes", p)
        X = h * (x + 1) / 2
        xx, yy, zz = X[None, None, :], X[None, :, None], X[:, None, None]
        u = (xx ** 2 + 2 * yy ** 2 + 3 * zz ** 2 + xx * yy * zz).reshape(-1).copy()
        J = np.full(n ** 3, h ** 3 / 8)
        rst = np.zeros((9, n ** 3)); rst[0] = rst[4] = rst[8] = 2 / h
        plan = Plan([p], [p], [0], [0], 0)
        plan.set_geometry(J, rst.reshape(-1))
        du = _t(u, gpu); dAu = torch.empty_like(du)
        plan.apply_stiffness_matrix(du, dAu)
        Au = dAu.cpu().numpy()
        assert abs((Au ** 2).sum() - case["Au_sq"]) <= 1e-12 * case["Au_sq"]
        assert abs(Au[0] - case["Au0"]) <= 1e-12 * np.abs(Au).max()


@pytest.mark.parametrize("level,deg,inc,qt", [(1, 1, 0, 0), (1, 3, 0, 0), (1, 3, 1, 0), (1, 7, 0, 0), (1, 7, 2, 0), (1, 2, 3, 0),
                                              (1, 4, 0, 1), (0, 11, 0, 0), (0, 15, 0, 0), (0, 18, 1, 0)])
def test_mass_galerkin_interp_dudr_parity(gpu, hiplib, oracle, level, deg, inc, qt):
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg, deg_quad_inc=inc, quad_type=qt)
    mp = M.SineMap(0.06)
    J, rst = m.geometry(mp)
    u = m.field(mp)
    plan = _plan(m, J, rst)
    du = _t(u, gpu)
    out = torch.full_like(du, float("nan"))
    plan.apply_mass_matrix(du, out)
    assert _rel(out.cpu().numpy(), oracle.apply_mass(m, J, u)) <= RTOL
    uq_ref = oracle.interpolate(m, u)
    duq = torch.full((m.local_nodes_quad,), float("nan"), dtype=torch.float64, device=gpu)
    plan.interpolate(du, duq)
    assert _rel(duq.cpu().numpy(), uq_ref) <= RTOL
    fq = np.cos(uq_ref)
    out2 = torch.full_like(du, float("nan"))
    plan.apply_galerkin_integral(_t(fq, gpu), out2)
    assert _rel(out2.cpu().numpy(), oracle.apply_galerkin(m, J, fq)) <= RTOL
    d = [torch.full_like(du, float("nan")) for _ in range(3)]
    plan.compute_dudr(du, *d)
    dref = oracle.compute_dudr(m, u)
    for i in range(3):
        assert _rel(d[i].cpu().numpy(), dref[i]) <= RTOL


@pytest.mark.parametrize("level,deg,inc,qt", [(1, 1, 0, 0), (1, 2, 0, 0), (1, 3, 0, 0), (1, 3, 2, 0), (1, 5, 1, 0), (1, 7, 0, 0), (1, 7, 1, 0),
                                              (1, 4, 0, 1), (1, 9, 0, 0), (0, 12, 0, 0), (0, 15, 0, 0), (0, 18, 0, 0), (0, 18, 1, 0)])
def test_weighted_inverse_mass_mij_parity(gpu, hiplib, oracle, level, deg, inc, qt):
    """d4est_quadrature_apply_fofufofvlilj / apply_inverse_mass_matrix, d4est_operators_apply_mij / invmij."""
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg, deg_quad_inc=inc, quad_type=qt)
    mp = M.SineMap(0.06)
    J, rst = m.geometry(mp)
    u = m.field(mp)
    plan = _plan(m, J, rst)
    du = _t(u, gpu)
    # weighted mass with the nonlinear-Poisson style coefficient f(u) = 1 + u^2 at the quadrature nodes
    uq = oracle.interpolate(m, u)
    coeff = 1.0 + uq * uq
    out = torch.full_like(du, float("nan"))
    plan.apply_weighted_mass_matrix(du, _t(coeff, gpu), out)
    assert _rel(out.cpu().numpy(), oracle.apply_weighted_mass(m, J, coeff, u)) <= RTOL
    # nodal mass applies
    for inverse in (False, True):
        o = torch.full_like(du, float("nan"))
        (plan.apply_invmij if inverse else plan.apply_mij)(du, o)
        assert _rel(o.cpu().numpy(), oracle.apply_mij(m, u, inverse)) <= RTOL
    # d4est_operators_apply_dij / _dij_transpose, one direction at a time
    import ctypes
    dp = ctypes.POINTER(ctypes.c_double)
    oracle.lib.oracle_apply_dij.argtypes = [dp, ctypes.c_int, ctypes.c_int, dp]
    oracle.lib.oracle_apply_dij_transpose.argtypes = [dp, ctypes.c_int, ctypes.c_int, dp]
    for direction in range(3):
        for tr in (False, True):
            o = torch.full_like(du, float("nan"))
            plan.apply_dij(du, direction, o, transpose=tr)
            ref = np.zeros(m.local_nodes)
            fn = oracle.lib.oracle_apply_dij_transpose if tr else oracle.lib.oracle_apply_dij
            for e in range(m.n_elements):
                s0, n3 = int(m.nodal_stride[e]), (int(m.deg[e]) + 1) ** 3
                src = np.ascontiguousarray(u[s0:s0 + n3]); dst = np.zeros(n3)
                fn(src.ctypes.data_as(dp), int(m.deg[e]), direction, dst.ctypes.data_as(dp))
                ref[s0:s0 + n3] = dst
            assert _rel(o.cpu().numpy(), ref) <= RTOL
    # d4est_operators_apply_slicer / _apply_lift on every face (bit-exact: pure gathers / scatters)
    oracle.lib.oracle_apply_slicer.argtypes = [dp, ctypes.c_int, ctypes.c_int, dp]
    oracle.lib.oracle_apply_lift.argtypes = [dp, ctypes.c_int, ctypes.c_int, dp]
    nf = plan.lib.d4est_hip_plan_face_nodes(plan.handle)
    fstride = np.concatenate([[0], np.cumsum((m.deg.astype(np.int64) + 1) ** 2)])
    assert nf == fstride[-1]
    for face in range(6):
        tf = torch.full((nf,), float("nan"), dtype=torch.float64, device=gpu)
        plan.apply_slicer(du, face, tf)
        back = torch.full_like(du, float("nan"))
        plan.apply_lift(tf, face, back)
        ref_f = np.zeros(nf); ref_b = np.zeros(m.local_nodes)
        for e in range(m.n_elements):
            s0, n3, n2 = int(m.nodal_stride[e]), (int(m.deg[e]) + 1) ** 3, (int(m.deg[e]) + 1) ** 2
            src = np.ascontiguousarray(u[s0:s0 + n3]); fo = np.zeros(n2); vo = np.zeros(n3)
            oracle.lib.oracle_apply_slicer(src.ctypes.data_as(dp), face, int(m.deg[e]), fo.ctypes.data_as(dp))
            oracle.lib.oracle_apply_lift(fo.ctypes.data_as(dp), int(m.deg[e]), face, vo.ctypes.data_as(dp))
            ref_f[fstride[e]:fstride[e] + n2] = fo
            ref_b[s0:s0 + n3] = vo
        assert np.array_equal(tf.cpu().numpy(), ref_f)
        assert np.array_equal(back.cpu().numpy(), ref_b)
    if inc == 0:
        # the inverse mass is Gauss-only in the reference; the tolerance carries the conditioning of V^-1 (grows with p)
        o = torch.full_like(du, float("nan"))
        plan.apply_inverse_mass_matrix(du, o)
        ref = oracle.apply_inverse_mass(m, J, u)
        assert _rel(o.cpu().numpy(), ref) <= 1e-11 * max(1, deg)
        if qt == 0:
            # M^-1 M u = u (the reference's identity test, d4est_test_operators / inverse mass definition)
            Mu = torch.empty_like(du); plan.apply_mass_matrix(du, Mu)
            back = torch.empty_like(du); plan.apply_inverse_mass_matrix(Mu, back)
            assert _rel(back.cpu().numpy(), u) <= 1e-10 * max(1, deg)


@pytest.mark.parametrize("level,deg,inc", [(1, 1, 0), (1, 3, 0), (2, 3, 2), (1, 5, 0), (1, 7, 0), (2, 7, 0), (1, 4, 0), (1, 6, 0),
                                           (1, 8, 0), (1, 9, 0), (1, 11, 0), (1, 12, 0), (0, 16, 0), (1, 15, 0)])
def test_affine_path_parity(gpu, hiplib, oracle, level, deg, inc):
    """Affine bricks: the engine rebuilds the metric from 6 numbers per element (tuning key 7 auto); same results as the
    general path and as the oracle.  A curved mesh must NOT be detected as affine; a mesh with ONE curved element neither."""
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg, deg_quad_inc=inc)
    J, rst = m.geometry(None)
    u = m.field()
    plan = _plan(m, J, rst)
    du = _t(u, gpu); a = torch.full_like(du, float("nan")); g = torch.full_like(du, float("nan"))
    plan.apply_stiffness_matrix(du, a)
    assert "affine" in plan.last_kernel()
    plan.set_tuning(7, 0)
    plan.apply_stiffness_matrix(du, g)
    assert "affine" not in plan.last_kernel()
    ref = oracle.apply_stiffness(m, J, rst, u)
    assert _rel(a.cpu().numpy(), ref) <= RTOL and _rel(g.cpu().numpy(), ref) <= RTOL
    plan.destroy()
    # perturb the geometry of one quadrature node of the last element: the bucket is no longer affine
    rst2 = rst.copy().reshape(9, -1)
    rst2[0, -1] *= 1.0 + 1e-9
    plan2 = _plan(m, J, rst2.reshape(-1))
    plan2.apply_stiffness_matrix(du, a)
    assert "affine" not in plan2.last_kernel()
    assert _rel(a.cpu().numpy(), oracle.apply_stiffness(m, J, rst2.reshape(-1), u)) <= RTOL
    plan2.destroy()


def test_edge_cases(gpu, hiplib, oracle):
    """empty plan, single element, ragged element count (not a multiple of elements-per-block)."""
    import torch
    from disco4est_amd import Plan, mesh as M
    empty = Plan([], [], [], [], 0)
    assert empty.local_nodes == 0
    empty.set_geometry(np.zeros(0), np.zeros(0))
    z = torch.zeros(0, dtype=torch.float64, device=gpu)
    empty.apply_stiffness_matrix(z, z.clone())
    torch.cuda.synchronize()
    for count in (1, 3, 5, 7):  # p=1: 16 elements per wavefront -> ragged tail
        m = M.BrickMesh(1, 1, first=0, count=count)
        J, rst = m.geometry(M.SineMap(0.05)); u = m.field()
        plan = _plan(m, J, rst)
        du = _t(u, gpu); dAu = torch.full_like(du, float("nan"))
        plan.apply_stiffness_matrix(du, dAu)
        assert _rel(dAu.cpu().numpy(), oracle.apply_stiffness(m, J, rst, u)) <= RTOL


def test_full_size_properties_config2(gpu, hiplib, oracle):
    """BASELINE config 2 (level 4, p = 7, 2 097 152 DoF): properties that need no oracle at size,
    plus an oracle spot-check on a sample of elements."""
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(4, 7)
    mp = M.SineMap(0.05)
    J, rst = m.geometry(mp)
    plan = _plan(m, J, rst)
    u = m.field(mp)
    du = _t(u, gpu)
    Au = torch.empty_like(du)
    plan.apply_stiffness_matrix(du, Au)
    scale = Au.abs().max().item()
    # constants are in the null space of every element matrix: sum over each element is ~0
    per_elem = Au.view(m.n_elements, -1).sum(dim=1).abs().max().item()
    assert per_elem <= 1e-10 * scale * 512
    ones = torch.ones_like(du); K1 = torch.empty_like(du)
    plan.apply_stiffness_matrix(ones, K1)
    assert K1.abs().max().item() <= 1e-11 * scale
    # symmetry: v.(K u) == u.(K v); linearity: K(a u + v) = a K u + K v
    v = _t(M.splitmix64_uniform(7, m.local_nodes), gpu)
    Kv = torch.empty_like(du); plan.apply_stiffness_matrix(v, Kv)
    s1, s2 = torch.dot(v, Au).item(), torch.dot(du, Kv).item()
    assert abs(s1 - s2) <= 1e-11 * max(abs(s1), abs(s2))
    lin = torch.empty_like(du); plan.apply_stiffness_matrix(2.5 * du + v, lin)
    assert (lin - (2.5 * Au + Kv)).abs().max().item() <= 1e-12 * scale * 4
    # positive semi-definite energy
    assert torch.dot(du, Au).item() > 0
    # determinism: two launches give identical bits
    Au2 = torch.empty_like(du); plan.apply_stiffness_matrix(du, Au2)
    assert torch.equal(Au, Au2)
    # oracle spot-check on 64 elements spread over the Morton curve
    got = Au.cpu().numpy()
    for e in range(0, m.n_elements, 64):
        sub = M.BrickMesh(4, 7, first=e, count=1)
        Je, rste = sub.geometry(mp)
        s = m.nodal_stride[e]
        ref = oracle.apply_stiffness(sub, Je, rste, np.ascontiguousarray(u[s:s + 512]))
        assert _rel(got[s:s + 512], ref) <= RTOL


@pytest.mark.parametrize("level,deg,inc", [(1, 3, 0), (2, "mixed", 1), (1, 7, 0), (1, 11, 0)])
def test_numerical_geometry_on_device(gpu, hiplib, oracle, level, deg, inc):
    """d4est_hip_plan_set_geometry_numerical (volume factors from the nodal coordinates, GEOM_COMPUTE_NUMERICAL) against the oracle's
    restatement of d4est_mesh.c:2637-2671: stiffness and mass with the device-made factors equal the oracle's with its own, 1e-12."""
    import torch
    from disco4est_amd import Plan, mesh as M
    if deg == "mixed":
        deg = 2 + (np.arange(8 ** level) % 4)
    m = M.BrickMesh(level, deg, deg_quad_inc=inc)
    mp = M.SineMap(0.05)
    xyz = m.nodal_coords(mp)
    Jn, rstn = oracle.geometry_numerical(m, xyz)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan.set_geometry_numerical(xyz)
    u = m.field(mp)
    du = torch.from_numpy(u).to(gpu)
    out = torch.empty_like(du)
    plan.apply_stiffness_matrix(du, out)
    ref = oracle.apply_stiffness(m, Jn, rstn, u)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
    plan.apply_mass_matrix(du, out)
    ref = oracle.apply_mass(m, Jn, u)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
    # device-resident coordinates take the same path
    plan2 = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0)
    plan2.set_geometry_numerical(torch.from_numpy(np.concatenate(xyz)).to(gpu))
    out2 = torch.empty_like(du)
    plan2.apply_mass_matrix(du, out2)
    assert torch.equal(out, out2)
    # the affine brick is still recognised (6 numbers per element instead of the per-node metric)
    m0 = M.BrickMesh(1, 7)
    p0 = Plan(m0.deg, m0.deg_quad, m0.nodal_stride, m0.quad_stride, 0)
    p0.set_geometry_numerical(m0.nodal_coords(None))
    x0 = torch.from_numpy(m0.field()).to(gpu); y0 = torch.empty_like(x0)
    p0.apply_stiffness_matrix(x0, y0)
    J0, rst0 = m0.geometry(None)
    ref0 = oracle.apply_stiffness(m0, J0, rst0, m0.field())
    assert np.abs(y0.cpu().numpy() - ref0).max() <= 1e-12 * np.abs(ref0).max()


@pytest.mark.parametrize("level,deg,count,label", [(5, 11, None, "config 3: 32 768 elements, 56.6 MDoF"),
                                                   (5, 15, 8192, "config 5's degree: 8192 elements, 33.6 MDoF")])
def test_full_size_properties_big_p(gpu, hiplib, oracle, level, deg, count, label):
    """BASELINE config 3 at full size and config 5's degree at bench size, general path (per-node metric streamed; factors generated
    on the device as in bench.py): the size-independent properties of the stiffness operator, two launches bit-identical, and the
    oracle on elements spread over the Morton curve."""
    import torch
    from disco4est_amd import Plan, mesh as M
    m = M.BrickMesh(level, deg, count=count)
    plan = Plan(m.deg, m.deg_quad, m.nodal_stride, m.quad_stride, 0, stream=torch.cuda.current_stream())
    plan.set_geometry_brick(np.ones(m.n_elements, dtype=np.int32), float(1 << level), [0.0, 1.0, 0.0, 1.0, 0.0, 1.0])
    plan.set_tuning(7, 0)
    n3 = (deg + 1) ** 3
    du = _t(M.splitmix64_uniform(11, m.local_nodes), gpu)
    Au = torch.empty_like(du)
    plan.apply_stiffness_matrix(du, Au)
    kern = plan.last_kernel()
    assert "affine" not in kern
    scale = Au.abs().max().item()
    assert scale > 0
    assert Au.view(m.n_elements, -1).sum(dim=1).abs().max().item() <= 1e-10 * scale * n3
    ones = torch.ones_like(du); K1 = torch.empty_like(du)
    plan.apply_stiffness_matrix(ones, K1)
    assert K1.abs().max().item() <= 1e-10 * scale
    del ones, K1
    v = _t(M.splitmix64_uniform(7, m.local_nodes), gpu)
    Kv = torch.empty_like(du); plan.apply_stiffness_matrix(v, Kv)
    s1, s2 = torch.dot(v, Au).item(), torch.dot(du, Kv).item()
    assert abs(s1 - s2) <= 1e-11 * max(abs(s1), abs(s2))
    lin = torch.empty_like(du); plan.apply_stiffness_matrix(2.5 * du + v, lin)
    assert (lin - (2.5 * Au + Kv)).abs().max().item() <= 1e-12 * scale * 8
    del lin, v, Kv
    assert torch.dot(du, Au).item() > 0
    Au2 = torch.empty_like(du); plan.apply_stiffness_matrix(du, Au2)
    assert torch.equal(Au, Au2)
    del Au2
    step = m.n_elements // 8
    for e in range(0, m.n_elements, step):
        sub = M.BrickMesh(level, deg, first=e, count=1)
        Je, rste = sub.geometry(None)
        s = int(m.nodal_stride[e])
        ue = du[s:s + n3].cpu().numpy()
        ref = oracle.apply_stiffness(sub, Je, rste, np.ascontiguousarray(ue))
        assert _rel(Au[s:s + n3].cpu().numpy(), ref) <= RTOL, (label, e, kern)
    plan.destroy()


@pytest.mark.parametrize("curved", [False, True])
def test_stiffness_multi_bucket_launch(gpu, hiplib, oracle, curved):
    """Mixed p = 1 ... 7 in one plan: all seven buckets run in ONE launch (stiffness_wave_eo_multi_kernel; streamed metric on the
    curved mesh, affine constants on the brick), held to the oracle element by element."""
    import torch
    from disco4est_amd import mesh as M
    deg = 1 + (np.arange(512) * 3) % 7
    m = M.BrickMesh(3, deg)
    mp = M.SineMap(0.05) if curved else None
    J, rst = m.geometry(mp)
    u = m.field(mp)
    ref = oracle.apply_stiffness(m, J, rst, u, nthreads=8)
    plan = _plan(m, J, rst)
    du = _t(u, gpu); dAu = torch.full_like(du, float("nan"))
    plan.apply_stiffness_matrix(du, dAu)
    name = plan.last_kernel()
    assert "stiffness_wave_eo_multi_kernel" in name and "7 buckets" in name, name
    assert ("affine" in name) == (not curved), name
    got = dAu.cpu().numpy()
    assert not np.isnan(got).any()
    for e in range(m.n_elements):
        s = m.nodal_stride[e]; n3 = (deg[e] + 1) ** 3
        assert _rel(got[s:s + n3], ref[s:s + n3]) <= 10 * RTOL, (e, deg[e])
    # the mass and weighted-mass applies of the same plan take the one-launch form too (mass_like_multi_kernel)
    out = torch.full_like(du, float("nan"))
    plan.apply_mass_matrix(du, out)
    refm = oracle.apply_mass(m, J, u)
    uq = oracle.interpolate(m, u)
    coeff = 1.0 + uq * uq
    outw = torch.full_like(du, float("nan"))
    plan.apply_weighted_mass_matrix(du, _t(coeff, gpu), outw)
    refw = oracle.apply_weighted_mass(m, J, coeff, u)
    gm, gw = out.cpu().numpy(), outw.cpu().numpy()
    for e in range(m.n_elements):
        s = m.nodal_stride[e]; n3 = (deg[e] + 1) ** 3
        assert _rel(gm[s:s + n3], refm[s:s + n3]) <= 10 * RTOL, (e, deg[e])
        assert _rel(gw[s:s + n3], refw[s:s + n3]) <= 10 * RTOL, (e, deg[e])
    # the general path forced on the brick: the streamed-metric form of the same launch
    if not curved:
        plan.set_tuning(7, 0)
        dAu2 = torch.full_like(du, float("nan"))
        plan.apply_stiffness_matrix(du, dAu2)
        assert "general" in plan.last_kernel()
        assert _rel(dAu2.cpu().numpy(), ref) <= RTOL
    plan.destroy()


@pytest.mark.parametrize("level,deg,tune", [(1, 5, ((1, 0), (0, 1))), (1, 7, ((1, 0), (0, 1))), (1, 8, ()), (1, 11, ()), (1, 13, ()), (0, 15, ()),
                                            (0, 17, ())])
def test_stream_mode_same_numbers(gpu, hiplib, oracle, level, deg, tune):
    """Tuning key 12 (non-temporal metric loads and A u stores on plans that do not fit the Infinity Cache; automatic by size) only changes
    cache hints: the NT instantiations of the volume kernels (the prefetching kernel of the large deg_quad <= 7 buckets, the multi-wave
    kernels, the matrix-core kernel) give the same bits as the plain ones, and the oracle's numbers."""
    import torch
    from disco4est_amd import mesh as M
    m = M.BrickMesh(level, deg)
    mp = M.SineMap(0.06)
    J, rst = m.geometry(mp)
    u = m.field(mp)
    ref = oracle.apply_stiffness(m, J, rst, u, nthreads=8)
    plan = _plan(m, J, rst)
    for k, v in tune:
        plan.set_tuning(k, v)
    du = _t(u, gpu)
    outs = []
    for key in (0, 1):
        plan.set_tuning(12, key)
        out = torch.full_like(du, float("nan"))
        plan.apply_stiffness_matrix(du, out)
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    assert _rel(outs[1].cpu().numpy(), ref) <= RTOL
    plan.destroy()
