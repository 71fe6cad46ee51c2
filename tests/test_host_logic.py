"""Host-side logic of the native path (no GPU): YAML schema, tags, mesh generators and markers, the
membrane-expression compiler, the AMG hierarchy builder."""
import os

import numpy as np
import pytest
import yaml

import knpemi_oracle as K
from cgx_hip import amg, fem
from cgx_hip import mesh as meshmod
from cgx_hip.parallel import extract_local, partition_mesh, stacked_cubes_local_mesh, stacked_squares_local_mesh, vertex_partition
from cgx_hip.problem import range_constructor
from parity_utils import CI_BASE, ci_config, make_oracle, make_problem


def test_yaml_schema_roundtrip(tmp_path):
    """A config written with the reference's keys (configs/tests/*.yaml) parses into the same attributes."""
    cfg = ci_config(N=8, steps=3)
    cfg.pop("quiet")
    f = tmp_path / "cfg.yaml"
    f.write_text(yaml.safe_dump(cfg))
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    cfgq = str(f)
    p = ProblemKNPEMI(cfgq)
    assert p.time_steps == 3 and p.dt.value == 2.5e-5
    assert p.intra_tags == (1,) and p.extra_tag == (2,) and p.gamma_tags == (4,) and p.stimulus_tags == (4,)
    assert abs(p.psi.value - 8.314 * 300 / 96485) < 1e-15 and p.C_M.value == 0.02
    assert p.g_syn_bar.value == 1e-9 and p.scale_stimulus is True and p.g_Na_leak.value == 0.3
    assert p.N_ions == 3 and [ion["name"] for ion in p.ion_list] == ["Na", "K", "Cl"]
    assert [ion["z"].value for ion in p.ion_list] == [1.0, 1.0, -1.0]
    assert p.solver_config["ksp_settings"]["ksp_rtol"] == 1e-9
    assert p.mesh.geometry.x.max() == pytest.approx(1e-6)


def test_range_tag_and_required_keys(tmp_path):
    yaml.add_constructor("!range", range_constructor, Loader=yaml.FullLoader)
    assert yaml.load("a: !range [2, 6]", Loader=yaml.FullLoader)["a"] == [2, 3, 4, 5]
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    for missing in ("solver", "dt", "ics_tags", "cell_tag_file"):
        cfg = ci_config(N=8)
        cfg.pop(missing)
        with pytest.raises(RuntimeError):
            ProblemKNPEMI(cfg)
    cfg = ci_config(N=8)
    cfg["cell_tag_file"] = "tissue.xdmf"
    with pytest.raises(RuntimeError):
        ProblemKNPEMI(cfg)


def test_ionic_model_tag_mismatch_raises():
    from CGx.KNPEMI.KNPEMIx_ionic_model import PassiveModel
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    p = ProblemKNPEMI(ci_config(N=8))
    p.set_initial_conditions()
    with pytest.raises(RuntimeError):
        p.init_ionic_models([PassiveModel(p, tags=(7,))])


@pytest.mark.parametrize("kind,N", [("square", 8), ("square", 12), ("cube", 4)])
def test_mesh_generators_and_gamma_match_oracle(kind, N):
    gen = meshmod.create_unit_square if kind == "square" else meshmod.create_unit_cube
    ogen = K.unit_square_mesh if kind == "square" else K.unit_cube_mesh
    c, t = gen(N)
    oc, ot = ogen(N)
    assert np.allclose(c, oc) and np.array_equal(t, ot)
    tags = meshmod.mark_subdomains_box(c, t)
    assert np.array_equal(tags, K.mark_subdomains(oc, ot))
    g, gt, _ = meshmod.gamma_integration_entities(t, tags, (1,), (2,))
    og = K.interface_facets(ot, np.where(tags == 1, 0, 1))
    assert np.array_equal(g, og) and np.all(gt == 4)
    # '+' side is intracellular for every facet
    assert np.all(tags[g[:, 0]] == 1) and np.all(tags[g[:, 2]] == 2)


def test_ci_mesh_sizes():
    c, t = meshmod.create_unit_square(32)
    tags = meshmod.mark_subdomains_box(c, t)
    g, _, _ = meshmod.gamma_integration_entities(t, tags, (1,), (2,))
    assert c.shape[0] == 1089 and t.shape[0] == 2048 and (tags == 1).sum() == 512 and g.shape[0] == 64


def test_facet_quadrature_matches_oracle():
    for dim in (2, 3):
        p, w = meshmod.facet_quadrature(dim, 10)
        op, ow = K.facet_quadrature(dim)
        assert np.allclose(p, op, atol=1e-15) and np.allclose(w, ow, atol=1e-15)


def test_membrane_program_matches_oracle_currents():
    """Compiler + bytecode semantics: interpret the compiled program of the CI mechanism set on the host
    and compare with the oracle's direct formulas at every membrane quadrature point."""
    p = make_problem(ci_config(N=16, steps=2))
    o = make_oracle(16)
    rng = np.random.default_rng(3)
    for side in range(2):
        for j in range(3):
            o.k[side][j] = o.k[side][j] * (1 + 0.1 * rng.random(o.n_v))
    o.phi_m = o.phi_m * (1 + 0.1 * rng.random(o.n_v))
    o.n, o.m, o.h = [a * (1 + 0.1 * rng.random(o.n_v)) for a in (o.n, o.m, o.h)]
    o.t = 3 * o.p.dt
    o.update_t_mod()
    p.t.value = o.t
    for m in p.ionic_models:
        if hasattr(m, "update_t_mod"):
            m.update_t_mod()
    Iq = o.channel_currents_q()
    ki = [o._at_q(o.k[0][j]) for j in range(3)]
    ke = [o._at_q(o.k[1][j]) for j in range(3)]
    aux = {"n": o._at_q(o.n), "m": o._at_q(o.m), "h": o._at_q(o.h)}
    spec = p.programs[0]
    assert spec.code.shape[1] == 4 and spec.code[:, 1].max() < 48
    out = fem.interpret_program(spec, ki, ke, o._at_q(o.phi_m), [aux[f.name] for f in p.aux_functions], [None] * 3)
    for j in range(3):
        assert np.max(np.abs(out[j] - Iq[j])) <= 1e-13 * np.max(np.abs(Iq[j]))


def test_expression_layer_semantics():
    m = type("M", (), {})()
    c = fem.Constant(m, 2.0)
    e = fem.conditional(fem.And(fem.gt(c, 1.0), fem.lt(c, 3.0)), c ** 3 + fem.sqrt(c) - fem.ln(c) / fem.exp(c), -1.0)
    assert bool(e) is True                               # UFL truthiness (f_NKCC1 quirk)
    v = fem.evaluate_numpy(e, {"x": [], "fields": {}})
    assert v == pytest.approx(8 + np.sqrt(2) - np.log(2) / np.exp(2))
    spec = fem.compile_program([e, 2 * c, c ** (3 / 2)], {})
    out = fem.interpret_program(spec, [0] * 3, [0] * 3, 0.0, [], [0] * 3)
    assert out[0] == pytest.approx(v) and out[1] == 4.0 and out[2] == pytest.approx(2 ** 1.5)
    c.value = 5.0                                        # constants are re-read at every upload
    out = fem.interpret_program(spec, [0] * 3, [0] * 3, 0.0, [], [0] * 3)
    assert out[0] == -1.0 and out[1] == 10.0


def test_passive_and_glial_models_compile():
    from CGx.KNPEMI.KNPEMIx_ionic_model import GlialCotransporters, KirNaKPumpModel
    p = make_problem(ci_config(N=8), models=lambda pr: [KirNaKPumpModel(pr), GlialCotransporters(pr)])
    o = K.make_square(8, models=[K.Model("kir_nak", (4,)), K.Model("glial_ct", (4,))])
    Iq = o.channel_currents_q()
    ki = [o._at_q(o.k[0][j]) for j in range(3)]
    ke = [o._at_q(o.k[1][j]) for j in range(3)]
    out = fem.interpret_program(p.programs[0], ki, ke, o._at_q(o.phi_m), [], [None] * 3)
    for j in range(3):
        assert np.max(np.abs(out[j] - Iq[j])) <= 1e-12 * np.max(np.abs(Iq[j]))


def test_amg_hierarchy_is_a_contraction():
    """Setup-side check: the hierarchy built for P, applied with the reference V-cycle, reduces the error of
    every elliptic block of P; GMRES on the CI problem converges in the reference's ~3 iterations."""
    o = make_oracle(32)
    P = o.assemble_P()
    h = amg.build_hierarchy(P)
    d = h.describe()
    assert len(d["rows"]) >= 2 and d["operator_complexity"] < 1.6
    M = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, 2)
    rng = np.random.default_rng(0)
    xt = rng.standard_normal(P.shape[0])
    b = P @ xt
    x = np.zeros_like(b)
    for _ in range(12):
        x = x + M(b - P @ x)
    # concentration blocks (SPD): fast contraction
    for f in range(3):
        assert np.linalg.norm((x - xt)[f::4]) <= 1e-3 * np.linalg.norm(xt[f::4])
    o2 = make_oracle(32)
    _, its = o2.run(3, solver="gmres", pc=lambda PP: K.pc_amg_vcycle(*(lambda hh: (hh.levels, hh.coarse_inv))(amg.build_hierarchy(PP)), 1, 1, 2), rtol=1e-9)
    assert max(its) <= 6


@pytest.mark.parametrize("fields,coarse", [((0, 1, 2, 3), 200), ((3,), 60), ((0, 1, 2), 150)])
def test_fused_cycle_restatement_equals_the_level_by_level_cycle(fields, coarse):
    """S = (I - c Dinv A) Pprol and Pt = A Dinv turn the V(1,1) / degree-1 cycle into two gathers per level; same operator."""
    o = make_oracle(16)
    P = o.assemble_P()
    Pm = P if len(fields) == 4 else amg.restrict_to_fields(P, fields)
    h = amg.build_hierarchy(Pm, coarse_size=coarse)
    assert len(h.levels) >= 2 and all(lv.S is not None for lv in h.levels[:-1])
    M0 = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, 1)
    M1 = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, 1, fused=True)
    r = np.random.default_rng(1).standard_normal(P.shape[0])
    for f in range(4):
        if f not in fields:
            r[f::4] = 0.0
    z0, z1 = M0(r), M1(r)
    act = np.isin(np.arange(P.shape[0]) % 4, fields)
    assert np.max(np.abs(z0[act] - z1[act])) <= 1e-12 * np.max(np.abs(z0[act]))
    # the fp32-stored variant the parity tests hand to the oracle keeps the same structure
    hf = amg.fp32_stored(h)
    assert hf.levels[0].Pt is not None and hf.levels[0].S.dtype == np.float64
    z2 = K.pc_amg_vcycle(hf.levels, hf.coarse_inv, 1, 1, 1, fused=True)(r)
    assert np.max(np.abs(z2[act] - z0[act])) <= 1e-5 * np.max(np.abs(z0[act]))


def test_aggregation_covers_all_nodes():
    o = make_oracle(16)
    P = o.assemble_P()
    S = amg.strength_graph(P, 0.08)
    agg, nagg = amg.aggregate(S)
    assert agg.min() == 0 and agg.max() == nagg - 1 and len(np.unique(agg)) == nagg
    # aggregates never mix fields or sides
    for a in range(0, nagg, max(1, nagg // 50)):
        members = np.nonzero(agg == a)[0]
        assert len(set(members % 4)) == 1
        assert len(set(o.lay.node_side[members // 4])) == 1


@pytest.mark.parametrize("size", [2, 3])
def test_partition_covers_mesh_once(size):
    c, t = meshmod.create_unit_square(12)
    tags = meshmod.mark_subdomains_box(c, t)
    g, gt, _ = meshmod.gamma_integration_entities(t, tags, (1,), (2,))
    seen_v = np.zeros(c.shape[0], int)
    seen_c = 0
    for r in range(size):
        lm = partition_mesh(c, t, tags, g, gt, size, r)
        seen_v[lm.l2g[:lm.n_vertices_owned]] += 1
        seen_c += lm.n_cells_owned
        # every cell touching an owned vertex is local
        owner = vertex_partition(c, size)
        assert lm.cells.shape[0] == int((owner[t] == r).any(axis=1).sum())
        assert np.allclose(lm.coords, c[lm.l2g])
        assert np.all(lm.ghost_owner != r)
    assert np.all(seen_v == 1) and seen_c == t.shape[0]


def test_stacked_generators_equal_global_mesh_on_one_rank():
    lm = stacked_squares_local_mesh(8, 1, 0)
    c, t = meshmod.create_unit_square(8)
    assert np.allclose(lm.coords, c) and lm.cells.shape == t.shape and lm.n_vertices_owned == c.shape[0]
    assert sorted(map(tuple, np.sort(lm.cells, axis=1).tolist())) == sorted(map(tuple, np.sort(t, axis=1).tolist()))
    lm3 = stacked_cubes_local_mesh(4, 1, 0)
    c3, t3 = meshmod.create_unit_cube(4)
    assert np.allclose(lm3.coords, c3) and lm3.cells.shape == t3.shape
    assert (lm3.cell_tags == 1).sum() == (meshmod.mark_subdomains_box(c3, t3) == 1).sum()


def test_tissue_lattice_generator_and_shared_program():
    """Tissue surrogate: one tag per cell, membrane tag = cell tag, identical mechanism lists on all tags
    compile to ONE membrane program; the stimulus area is integrated once."""
    import numpy as np
    from cgx_hip import mesh as M
    from parity_utils import make_problem, tissue_config
    coords, cells, tags, ft, desc = M.load_mesh("tissue3d_8_2.xdmf", "", 1.0)
    assert ft == "intra" and "tissue" in desc
    assert sorted(np.unique(tags).tolist()) == list(range(1, 10))
    # every cell is a (B-2)^3 block of voxels, 6 tets each
    counts = np.bincount(tags)[2:]
    assert np.all(counts == 6 * 2 ** 3)
    gamma, gtags, _ = M.gamma_integration_entities(cells, tags, tuple(range(2, 10)), (1,), ft)
    assert np.array_equal(gtags, tags[gamma[:, 0]]) and np.all(tags[gamma[:, 2]] == 1)
    p = make_problem(tissue_config(2, 16, 2, steps=1), "ci")
    assert len(p.gamma_tags) == 4 and len(p.programs) == 1
    assert set(p.tag_program.values()) == {0}
    assert len(p._stimulus_area_cache) == 1


def test_ion_injection_site_and_sources():
    """source_terms: ion_injection (mixed_dim_problem.py:496-541, KNPEMIx_problem.py:200-218)."""
    import numpy as np
    from parity_utils import make_problem, tissue_config
    cfg = tissue_config(3, 10, 2, steps=1, stimulus=False)
    cfg["source_terms"] = "ion_injection"
    p = make_problem(cfg, "passive")
    h = 1e-6 / 10
    # the cube of half-width (x_max - x_min)/10 around the centre: 2x2x2 voxels of 6 tets
    assert len(p.injection_cells) == 48
    assert abs(p.injection_volume - 8 * h ** 3) <= 1e-12 * h ** 3
    f = p.ion_list[1]["f_e"].numpy()
    assert np.count_nonzero(f) == 27 and np.allclose(f[f != 0], 5e-9 / float(p.F.value) / p.injection_volume)
    assert np.array_equal(f, p.ion_list[2]["f_e"].numpy())
    assert not hasattr(p.ion_list[0]["f_e"], "numpy")      # Na keeps the zero Constant


def test_dirichlet_bc_data_outside_mms():
    """KNPEMIx_problem.py:135-160: every field pinned on the exterior boundary, values captured at construction
    (class defaults), intra fields only where an intracellular node exists."""
    import numpy as np
    from parity_utils import ci_config, make_problem
    cfg = ci_config(N=8, steps=1)
    cfg["dirichlet_bcs"] = True
    p = make_problem(cfg)
    assert p.dirichlet_bcs and len(p.bcs) == 8
    assert len(p.bc_vertices) == 32
    vals = {(side, f): v[0] for side, f, _, v in p.bcs}
    assert vals[("extra", 0)] == 145.0 and vals[("extra", 1)] == 3.0 and vals[("extra", 2)] == 134.0 and vals[("extra", 3)] == 0.0
    assert vals[("intra", 0)] == 10.0 and vals[("intra", 3)] == -0.07


def test_membrane_programs_compile_to_native_code_for_gfx950():
    """csrc/knp_jit.cpp: the bytecode of the CI mechanism set (and of the glial set) is emitted as HIP source in front of
    the facet kernel and compiles with hiprtc for gfx950 -- no GPU needed for this check."""
    import ctypes as C
    import numpy as np
    from cgx_hip import _lib
    from parity_utils import ci_config, make_problem, two_cell_config
    lib = _lib.load()
    progs = list(make_problem(ci_config(N=8, steps=1)).programs.values())
    assert progs and progs[0].code.shape[0] > 50
    for spec in progs:
        code = np.ascontiguousarray(spec.code, dtype=np.int32)
        log = C.create_string_buffer(4096)
        rc = lib.knp_jit_compile_check(code.ctypes.data_as(C.POINTER(C.c_int32)), code.shape[0], b"gfx950", log, 4096)
        assert rc == 0, log.value.decode()
        assert log.value.decode().startswith("ok")
    # an instruction the code generator does not know is reported, not compiled
    bad = np.array([[9999, 0, 0, 0]], dtype=np.int32)
    log = C.create_string_buffer(4096)
    assert lib.knp_jit_compile_check(bad.ctypes.data_as(C.POINTER(C.c_int32)), 1, b"gfx950", log, 4096) != 0


def test_generated_membrane_code_hoists_constant_only_subexpressions(tmp_path, monkeypatch):
    """csrc/knp_jit.cpp: instructions of a membrane program whose operands trace back to its constant table only (psi / z_k,
    exp(-t / a_syn): a division or an exponential each, identical at all 36 quadrature points of a facet) are emitted into a prologue
    ``knp_prog_<id>_pre`` that runs once per thread; the per-point function reads their results from ``U``.  The number of auxiliary
    fields and the coordinate axes a program reads become compile-time constants of the generated source.  (That both versions compute
    the same currents is the GPU test test_runtime_compiled_membrane_programs_match_interpreter.)"""
    import ctypes as C
    import numpy as np
    from cgx_hip import _lib
    from cgx_hip.fem import OPS
    lib = _lib.load()
    monkeypatch.setenv("KNP_JIT_DUMP", str(tmp_path))
    # r0 = C[0]; r1 = C[1]; r2 = r0 / r1 (constants only); r3 = exp(r2) (constants only); r4 = ki[0]; r5 = aux[1]; r4 = r4 * r3;
    # r4 = r4 / r5 (per point); r6 = xq[1]; r4 = r4 + r6; I[0] += r4
    code = np.array([[OPS["CONST"], 0, 0, 0], [OPS["CONST"], 1, 1, 0], [OPS["DIV"], 2, 0, 1], [OPS["EXP"], 3, 2, 0], [OPS["KI"], 4, 0, 0],
                     [OPS["AUX"], 5, 1, 0], [OPS["MUL"], 4, 4, 3], [OPS["DIV"], 4, 4, 5], [OPS["X"], 6, 1, 0], [OPS["ADD"], 4, 4, 6],
                     [OPS["OUT"], 0, 0, 4]], dtype=np.int32)
    log = C.create_string_buffer(4096)
    rc = lib.knp_jit_compile_check(code.ctypes.data_as(C.POINTER(C.c_int32)), code.shape[0], b"gfx950", log, 4096)
    assert rc == 0, log.value.decode()
    src = (tmp_path / "knp_gamma_jit.hip").read_text()
    pre = src[src.index("void knp_prog_0_pre("):src.index("void knp_prog_0(")]
    main = src[src.index("void knp_prog_0("):src.index("void knp_jit_pre(")]
    assert "r2 = r0 / r1; U[0] = r2;" in pre and "r3 = exp(r2); U[1] = r3;" in pre
    assert "exp(" not in main and "r0 / r1" not in main and "r3 = U[1];" in main and "r4 = r4 / r5;" in main
    assert "#define KNP_JIT_NAUX 2" in src and "#define KNP_JIT_XMASK 2" in src
    assert (tmp_path / "knp_gamma_jit.hsaco").stat().st_size > 1000


def test_device_builder_restricts_the_fields_itself():
    """``amg_gpu.build_hierarchy(P, fields=...)``: the hierarchy of a field class of P built from the unrestricted matrix (the restriction
    runs on the device) is the one built from ``amg.restrict_to_fields(P, fields)``."""
    import numpy as np
    import knpemi_oracle as K
    from cgx_hip import amg, amg_gpu
    P = K.make_cube(6, models=K.CI_MODELS()).assemble_P()
    for fields, nf in (((0, 1, 2), (4, (0, 1, 2))), ((3,), None)):
        kw = dict(theta=0.08, coarse_size=60, device="cpu", node_fields=nf, agg_distance=[2, 1])
        h0 = amg_gpu.build_hierarchy(amg.restrict_to_fields(P, fields), **kw)
        h1 = amg_gpu.build_hierarchy(P, fields=fields, **kw)
        assert h0.describe() == h1.describe() and len(h0.levels) >= 2
        assert (h0.levels[0].A != h1.levels[0].A).nnz == 0
        for a, b in zip(h0.levels, h1.levels):
            assert abs(a.A - b.A).max() == 0.0 and np.array_equal(a.dinv, b.dinv)


def test_device_amg_setup_reproduces_the_host_setup():
    """cgx_hip/amg_gpu.py (torch sparse products; run here on CPU tensors) builds the hierarchy of cgx_hip/amg.py: same
    aggregates and level sizes, operators equal to rounding."""
    import numpy as np
    import knpemi_oracle as K
    from cgx_hip import amg, amg_gpu
    o = K.make_cube(8, models=K.CI_MODELS())
    P = o.assemble_P()
    for fields in ((0, 1, 2), (3,), (0, 1, 2, 3)):
        Pm = amg.restrict_to_fields(P, fields) if len(fields) < 4 else P
        h0 = amg.build_hierarchy(Pm, theta=0.08, coarse_size=100)
        h1 = amg_gpu.build_hierarchy(Pm, theta=0.08, coarse_size=100, device="cpu")
        assert h0.describe()["rows"] == h1.describe()["rows"] and len(h0.levels) >= 2
        for a, b in zip(h0.levels, h1.levels):
            assert abs(a.A - b.A).max() <= 1e-12 * abs(a.A).max()
            assert abs(a.lambda_max - b.lambda_max) <= 1e-10 * a.lambda_max
            assert np.abs(a.dinv - b.dinv).max() <= 1e-12 * np.abs(a.dinv).max()
            if a.P is not None:
                assert abs(a.P - b.P).max() <= 1e-12 and abs(a.R - b.R).max() <= 1e-12
        assert np.abs(h0.coarse_inv - h1.coarse_inv).max() <= 1e-8 * np.abs(h0.coarse_inv).max()


def test_distance_one_aggregation_host_and_device_builders_agree():
    """``agg_distance`` per level (SolverKNPEMI.ion_agg_distance: "2,1" in 3D -- distance-2 aggregates on the finest level, distance 1
    below): roots from an independent set of the strength graph itself give smaller aggregates and more levels; both builders produce
    the same hierarchies, and the V(1,1) cycle converges faster per cycle than with distance 2 everywhere on the thin-sheet lattice."""
    import numpy as np
    import knpemi_oracle as K
    from cgx_hip import amg, amg_gpu, mesh as M
    name = "tissue3d_9_2_w1.xdmf"
    coords, cells, tags, ft, _ = M.load_mesh(name, name, 1e-6)
    intra = tuple(int(t) for t in np.unique(tags) if t != 1)
    gam, gt, _ = M.gamma_integration_entities(cells, tags, intra, (1,), "intra")
    o = K.OracleKNPEMI(coords / 1e-6, cells, tags, intra_tags=intra, extra_tag=1, gamma=gam, gamma_tag=gt, models=[K.Model("passive", intra)],
                       mesh_conversion_factor=1e-6)
    Pk = amg.restrict_to_fields(o.assemble_P(), (0, 1, 2))
    kw = dict(theta=0.08, coarse_size=60, node_fields=(4, (0, 1, 2)))
    h2 = amg.build_hierarchy(Pk, agg_distance=[2], **kw)
    h1 = amg.build_hierarchy(Pk, agg_distance=[1], **kw)
    g1 = amg_gpu.build_hierarchy(Pk, agg_distance=[1], device="cpu", **kw)
    assert h1.describe()["rows"] == g1.describe()["rows"]
    kw21 = dict(theta=0.08, coarse_size=30, node_fields=(4, (0, 1, 2)), agg_distance=[2, 1])
    h21, g21 = amg.build_hierarchy(Pk, **kw21), amg_gpu.build_hierarchy(Pk, device="cpu", **kw21)
    assert h21.describe()["rows"] == g21.describe()["rows"] and h21.levels[1].A.shape[0] == h2.levels[1].A.shape[0]
    assert len(h1.levels) >= len(h2.levels) and h1.levels[1].A.shape[0] > h2.levels[1].A.shape[0]      # smaller aggregates
    for a, b in zip(h1.levels, g1.levels):
        assert abs(a.A - b.A).max() <= 1e-12 * abs(a.A).max()
        if a.P is not None:
            assert abs(a.P - b.P).max() <= 1e-12
    # convergence factor of the stationary iteration e <- (I - V A) e on the ion block
    n = Pk.shape[0]
    ion = np.setdiff1d(np.arange(n), np.arange(3, n, 4))

    def factor(h):
        V = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, 1)
        e = np.zeros(n)
        e[ion] = np.random.default_rng(0).standard_normal(ion.size)
        r = 1.0
        for _ in range(12):
            e2 = e - V(Pk @ e)
            e2[3::4] = 0.0
            r = np.linalg.norm(e2) / np.linalg.norm(e)
            e = e2 / np.linalg.norm(e2)
        return r
    f1, f2 = factor(h1), factor(h2)
    assert f1 < f2 and f1 < 0.6, (f1, f2)


def test_node_synchronised_aggregation_shares_patterns_between_fields():
    """``node_fields`` (amg.build_hierarchy): the ion fields are aggregated once, on the node graph of the first one, so
    coarse unknowns are numbered nf*aggregate+k and every operator of the hierarchy has the same sparsity pattern in all
    fields and no coupling between them (the ion blocks of P are uncoupled, reference KNPEMI/ProblemKNPEMI.py:560-620).
    Host and device builders agree."""
    import numpy as np
    import knpemi_oracle as K
    from cgx_hip import amg, amg_gpu
    o = K.make_cube(8, models=K.CI_MODELS())
    Pm = amg.restrict_to_fields(o.assemble_P(), (0, 1, 2))
    h0 = amg.build_hierarchy(Pm, theta=0.08, coarse_size=100, node_fields=(4, (0, 1, 2)))
    h1 = amg_gpu.build_hierarchy(Pm, theta=0.08, coarse_size=100, device="cpu", node_fields=(4, (0, 1, 2)))
    assert h0.node_fields == 3 and h1.node_fields == 3 and len(h0.levels) >= 3
    assert h0.describe()["rows"] == h1.describe()["rows"]
    for li, (a, b) in enumerate(zip(h0.levels, h1.levels)):
        assert abs(a.A - b.A).max() <= 1e-12 * abs(a.A).max()
        if a.P is not None:
            assert abs(a.P - b.P).max() <= 1e-12
        stride = 4 if li == 0 else 3
        mats = [(a.A, stride, (0, 1, 2))] + ([(a.P, 3, (0, 1, 2)), (a.S, 3, (0, 1, 2))] if a.P is not None else [])
        for M, cstride, cfields in mats:
            M = M.tocsr()
            pats = []
            for f, cf in zip((0, 1, 2), cfields):
                rows = M[f::stride]
                blk = rows[:, cf::cstride].tocsr()
                blk.sort_indices()
                assert blk.nnz == rows.nnz, "an ion row couples to another field"
                pats.append((blk.indptr.copy(), blk.indices.copy()))
            for pp in pats[1:]:
                assert np.array_equal(pp[0], pats[0][0]) and np.array_equal(pp[1], pats[0][1])


def test_node_synchronised_aggregation_declines_an_indefinite_potential_block():
    """P of the MMS problem on the 32x32 square (tests/golden/mms_P_square32.npz, assembled by the HIP path and dumped with
    tests/devtools/dump_mms.py): non-dimensional constants make the membrane mass of the potential block (minus sign in the
    reference's form, KNPEMIx_problem.py:657-744) larger than its stiffness, the diagonal turns negative on membrane nodes.
    Following the ion aggregates there gives a V-cycle with eigenvalues of both signs (GMRES then stalls: 2000 iterations
    on the GPU); both builders must fall back to the unsynchronised aggregation for such a matrix."""
    import os
    import numpy as np
    import scipy.sparse as sp
    from cgx_hip import amg, amg_gpu
    P = sp.load_npz(os.path.join(os.path.dirname(__file__), "golden", "mms_P_square32.npz")).tocsr()
    assert P.diagonal()[3::4].min() < 0 < P.diagonal()[0::4].min()
    h0 = amg.build_hierarchy(P)
    for h in (amg.build_hierarchy(P, node_fields=(4, (0, 1, 2, 3))), amg_gpu.build_hierarchy(P, node_fields=(4, (0, 1, 2, 3)), device="cpu")):
        assert h.node_fields == 0 and h.describe()["rows"] == h0.describe()["rows"]
    # the ion fields alone are fine
    assert amg.build_hierarchy(amg.restrict_to_fields(P, (0, 1, 2)), node_fields=(4, (0, 1, 2))).node_fields == 3


def test_decoupled_unknowns_are_split_off_the_hierarchy(monkeypatch):
    """Tissue lattice (216 cells, one tag each): after two levels every intracellular cell is one aggregate per field, a 1x1
    block of the level operator.  ``split_decoupled`` keeps those out of the coarser levels (exact degree-1 smoothing:
    inverse diagonal divided by the Chebyshev coefficient, zero prolongator row) and injects the rest into a last level that
    is small enough for the dense inverse; the V-cycle stays the exact inverse on the split-off unknowns.  Host and device
    builders agree."""
    import numpy as np
    import knpemi_oracle as K
    from cgx_hip import amg, amg_gpu, mesh as M
    coords, cells, tags, ft, _ = M.load_mesh("tissue3d_13_3_w1.xdmf", "tissue3d_13_3_w1.xdmf", 1e-6)
    intra = tuple(int(t) for t in np.unique(tags) if t != 1)
    gam, gt, _ = M.gamma_integration_entities(cells, tags, intra, (1,), "intra")
    o = K.OracleKNPEMI(coords / 1e-6, cells, tags, intra_tags=intra, extra_tag=1, gamma=gam, gamma_tag=gt,
                       models=[K.Model("passive", intra)], mesh_conversion_factor=1e-6)
    Pk = amg.restrict_to_fields(o.assemble_P(), (0, 1, 2))
    monkeypatch.setattr(amg, "DENSE_LIMIT", 70)          # (the real limit, 6000, is reached by lattices of a few thousand cells)
    kw = dict(theta=0.08, coarse_size=60, node_fields=(4, (0, 1, 2)))
    h0 = amg.build_hierarchy(Pk, split_decoupled=False, **kw)
    h1 = amg.build_hierarchy(Pk, **kw)
    h2 = amg_gpu.build_hierarchy(Pk, device="cpu", **kw)
    ncell = len(intra)
    assert h1.describe()["rows"] == h2.describe()["rows"] and h1.node_fields == 3
    assert h1.levels[-1].A.shape[0] <= 60 < h0.levels[-1].A.shape[0] and h1.coarse_inv is not None
    for a, b in zip(h1.levels, h2.levels):
        assert abs(a.A - b.A).max() <= 1e-12 * abs(a.A).max() and np.abs(a.dinv - b.dinv).max() <= 1e-12 * np.abs(a.dinv).max()
        if a.P is not None:
            assert abs(a.P - b.P).max() <= 1e-12 and abs(a.S - b.S).max() <= 1e-12
    # the level on which the cells have collapsed: 3 * ncell decoupled rows with zero prolongator rows and the scaled inverse diagonal
    found = False
    for lv in h1.levels[:-1]:
        A = lv.A.tocsr()
        iso = amg._decoupled_rows(A, A.diagonal())
        if iso.sum() >= 3 * ncell:
            found = True
            assert np.diff(lv.P.tocsr().indptr)[iso].max() == 0 and np.diff(lv.S.tocsr().indptr)[iso].max() == 0
            c = amg.cheby_first_coefficient(lv.lambda_max)
            assert np.allclose(lv.dinv[iso] * A.diagonal()[iso] * c, 1.0, rtol=1e-13)
    assert found
    # same V-cycle quality: GMRES on P itself
    import scipy.sparse.linalg as spla
    n = Pk.shape[0]
    act = Pk.diagonal() != 0
    b = np.where(act, np.random.default_rng(0).standard_normal(n), 0.0)
    its = []
    for h in (h0, h1):
        V = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, 1)
        x, k, _ = K.gmres_left(Pk, b, np.zeros(n), V, rtol=1e-8, max_it=200)
        its.append(k)
    assert its[1] <= its[0] + 1, its


def test_decoupled_split_only_with_a_degree_one_smoother():
    """the split divides the inverse diagonal of 1x1 blocks by the degree-1 Chebyshev coefficient, which is exact only
    when every smoothing step is a damped Jacobi step.  With ``amg_cheby_degree`` 2 the builders must not split (the momentum term
    of the second step overshoots a zero residual): one V(1,1) application on a Laplacian plus 4000 decoupled 1x1 blocks must
    reproduce b/d on the decoupled unknowns for degree 1 (split) and degree 2 (no split) alike."""
    import numpy as np
    import scipy.sparse as sp
    import knpemi_oracle as K
    from cgx_hip import amg, amg_gpu
    m = 40
    T = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))
    Lap = (sp.kron(sp.identity(m), T) + sp.kron(T, sp.identity(m)) + 1e-3 * sp.identity(m * m)).tocsr()
    d1 = 0.5 + np.random.default_rng(3).random(4000)
    P = sp.block_diag([Lap, sp.diags(d1)]).tocsr()
    b = np.random.default_rng(4).standard_normal(P.shape[0])
    for build in (amg.build_hierarchy, lambda M, **kw: amg_gpu.build_hierarchy(M, device="cpu", **kw)):
        for deg in (1, 2):
            h = build(P, theta=0.08, coarse_size=100, smoother_degree=deg)
            lv0 = h.levels[0]
            split = np.diff(lv0.P.tocsr().indptr)[m * m:].max() == 0
            assert split == (deg == 1)
            z = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, deg)(b)
            err = np.abs(z[m * m:] * d1 - b[m * m:]).max() / np.abs(b[m * m:]).max()
            assert err <= (1e-12 if deg == 1 else 0.2), (deg, err)      # degree 2 without split: smoothed singletons, no overshoot of 0.28
        # the overshoot the finding describes, for the record: splitting under a degree-2 smoother is wrong
        h_bad = amg.build_hierarchy(P, theta=0.08, coarse_size=100, smoother_degree=1)
        zb = K.pc_amg_vcycle(h_bad.levels, h_bad.coarse_inv, 1, 1, 2)(b)
        assert np.abs(zb[m * m:] * d1 - b[m * m:]).max() / np.abs(b[m * m:]).max() > 0.2


def test_recursive_coordinate_bisection_partition():
    """General meshes are cut by recursive coordinate bisection: balanced, deterministic, compact (2x2x2 blocks on a
    cube for 8 ranks -> far fewer cut edges than 8 slabs)."""
    import numpy as np
    from cgx_hip import mesh as M
    from cgx_hip.parallel import vertex_partition
    coords, cells = M.create_unit_cube(8)
    for size in (2, 3, 5, 8):
        o = vertex_partition(coords, size)
        cnt = np.bincount(o, minlength=size)
        assert cnt.max() - cnt.min() <= size and np.array_equal(o, vertex_partition(coords, size))
    def cut_edges(owner):
        e = np.concatenate([cells[:, [a, b]] for a in range(4) for b in range(a + 1, 4)])
        return int((owner[e[:, 0]] != owner[e[:, 1]]).sum())
    assert cut_edges(vertex_partition(coords, 8, "rcb")) < 0.6 * cut_edges(vertex_partition(coords, 8, "slab"))
    # weighted cuts: vertices in one corner count double (membrane vertices carry two nodes) -> the WEIGHT is balanced
    w = np.where((coords < 0.4).all(axis=1), 2.0, 1.0)
    for size in (2, 4, 7):
        for method in ("rcb", "slab"):
            o = vertex_partition(coords, size, method, weights=w)
            load = np.bincount(o, weights=w, minlength=size)
            assert load.max() - load.min() <= 2.0 * size + 2 and (np.bincount(o, minlength=size) > 0).all()
            cnt = np.bincount(o, minlength=size)
            assert cnt.max() - cnt.min() > size, "weights must move the cuts"


def test_point_location_and_p1_weights():
    """probe points (reference scifem.evaluate_function): containing cell + barycentric weights reproduce linear functions"""
    from cgx_hip import mesh as meshmod
    from cgx_hip.output import _barycentric
    for gen, N in ((meshmod.create_unit_square, 6), (meshmod.create_unit_cube, 3)):
        coords, cells = gen(N)
        d = coords.shape[1]
        rng = np.random.default_rng(5)
        pts = np.vstack([rng.random((20, d)), coords[[0, len(coords) // 2, -1]], coords[cells[3]].mean(axis=0)[None]])
        c, w = _barycentric(coords, cells, pts)
        assert (c >= 0).all()
        lin = coords @ np.arange(1, d + 1) + 0.5
        val = (lin[cells[c]] * w).sum(axis=1)
        assert np.allclose(val, pts @ np.arange(1, d + 1) + 0.5, rtol=1e-12, atol=1e-12)
        c2, _ = _barycentric(coords, cells, np.full((1, d), 1.5))
        assert c2[0] == -1


def test_point_evaluation_key_is_parsed_and_scaled():
    import torch
    cfg = ci_config(N=8, steps=1)
    cfg["point_evaluation"] = {"ics_points": [[0.5, 0.5]], "ecs_points": [[0.1, 0.1], [0.9, 0.2]], "gamma_points": [[0.25, 0.5]]}
    p = make_problem(cfg)
    assert p.point_evaluation and p.ics_points.shape == (1, 2) and p.ecs_points.shape == (2, 2)
    assert np.allclose(p.gamma_points, [[0.25e-6, 0.5e-6]])
    cfg.pop("point_evaluation")
    assert not make_problem(cfg).point_evaluation


def test_main_cli_interface():
    """same command line as the reference's main.py (--config, --view)"""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PYTHONPATH=os.path.join(root, "knp-emi-cgx_amd"))
    out = subprocess.run([sys.executable, "-m", "CGx.KNPEMI.main", "--help"], capture_output=True, text=True, env=env, cwd=root)
    assert out.returncode == 0 and "--config" in out.stdout and "--view" in out.stdout


@pytest.mark.parametrize("fields", [(0, 1, 2, 3), (0, 1, 2), (3,)])
def test_hierarchy_setup_invariants(fields):
    """Independent structural checks of the AMG setup (the V-cycle parity tests only check its APPLICATION): the restrictor is the
    transposed prolongator, every coarse operator is the Galerkin product, aggregates cover exactly the active unknowns, the
    prolongator reproduces constants wherever the operator annihilates them, and S is the post-smoothed prolongator."""
    import scipy.sparse as sp
    o = make_oracle(16)
    P = o.assemble_P()
    A0 = P if len(fields) == 4 else amg.restrict_to_fields(P, fields)
    h = amg.build_hierarchy(A0, coarse_size=60 if len(fields) > 1 else 20)
    assert len(h.levels) >= 3
    for l, lv in enumerate(h.levels[:-1]):
        A, Pm, R, S = lv.A, lv.P, lv.R, lv.S
        nxt = h.levels[l + 1].A
        assert abs(R - Pm.T).max() == 0.0
        G = (R @ (A @ Pm)).tocsr()
        assert abs(G - nxt).max() <= 1e-12 * abs(nxt).max()
        active = A.diagonal() != 0.0
        dec = amg._decoupled_rows(A.tocsr(), A.diagonal())               # 1x1 blocks: solved by the smoother, not interpolated
        if dec.any():
            assert np.allclose(lv.dinv[dec] * A.diagonal()[dec] * amg.cheby_first_coefficient(lv.lambda_max), 1.0, rtol=1e-13)
        active = active & ~dec
        rows_with_P = np.diff(Pm.indptr) > 0
        assert np.array_equal(rows_with_P, active)                       # inactive fields get no interpolation
        assert (np.diff(nxt.indptr) > 0).all()                           # every coarse unknown is coupled
        c = amg.cheby_first_coefficient(lv.lambda_max)
        S_ref = (Pm - sp.diags(c * lv.dinv) @ (A @ Pm)).tocsr()
        assert abs(S - S_ref).max() <= 1e-13 * abs(S_ref).max()
        # smoothed aggregation interpolates constants exactly on rows whose FILTERED operator has zero row sum; here: check the
        # weaker, filter-independent statement that P 1 is within [0.5, 1.5] on active rows (partition of unity up to smoothing)
        one = Pm @ np.ones(Pm.shape[1])
        assert one[active].min() > 0.5 and one[active].max() < 1.5
    # the dense coarse solve is a pseudo-inverse of the last operator
    Ac = h.levels[-1].A.toarray()
    assert np.abs(Ac @ h.coarse_inv @ Ac - Ac).max() <= 1e-8 * np.abs(Ac).max()
    # the cycle contracts on the range of the hierarchy's fields
    M = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, 1)
    rng = np.random.default_rng(2)
    xt = rng.standard_normal(P.shape[0])
    for f in range(4):
        if f not in fields:
            xt[f::4] = 0.0
    b = A0 @ xt
    x = np.zeros_like(b)
    for _ in range(25):
        x = x + M(b - A0 @ x)
    for f in fields:
        if f < 3:
            assert np.linalg.norm((x - xt)[f::4]) <= 1e-4 * np.linalg.norm(xt[f::4])


def test_independent_set_elimination_level_is_an_exact_block_factorisation():
    """The coupled potential block of a tissue mesh reaches a level of 'one unknown per biological cell + extracellular aggregates':
    the cells couple to the aggregates but not to each other.  ``build_hierarchy`` eliminates such an independent set exactly
    (ideal interpolation, smoother exact on the set) instead of aggregating it.  On a model matrix [[D, B], [B^T, Ae]] with 400
    mutually uncoupled unknowns and 30 coupled ones, the two-level cycle with the dense coarse solve IS the inverse -- in the
    level-by-level order and in the fused order (Rt / U products, S) alike; host and device builder agree."""
    import numpy as np
    import scipy.sparse as sp
    import knpemi_oracle as K
    from cgx_hip import amg, amg_gpu
    rng = np.random.default_rng(5)
    nF, nC = 400, 30
    B = sp.random(nF, nC, density=0.08, random_state=7, data_rvs=lambda k: -rng.random(k)).tocsr()
    T = sp.diags([-1.0, 2.2, -1.0], [-1, 0, 1], shape=(nC, nC))
    Ae = (T + sp.diags(np.asarray(abs(B).sum(axis=0)).ravel())).tocsr()
    D = sp.diags(np.asarray(abs(B).sum(axis=1)).ravel() + 0.05)
    A = sp.bmat([[D, B], [B.T, Ae]], format="csr")
    F = amg.low_degree_independent_set(A, np.ones(nF + nC, dtype=bool))
    assert F[:nF].all() and not F[nF:].any()                      # the low-degree rows, and only they
    b = rng.standard_normal(nF + nC)
    x = np.linalg.solve(A.toarray(), b)
    for build in (amg.build_hierarchy, lambda M, **kw: amg_gpu.build_hierarchy(M, device="cpu", **kw)):
        h = build(A, coarse_size=40)
        assert [lv.A.shape[0] for lv in h.levels] == [nF + nC, nC] and h.coarse_inv is not None
        assert np.abs(h.levels[0].dinv[nF:]).max() == 0.0          # no smoothing on the remainder
        for fused in (False, True):
            z = K.pc_amg_vcycle(h.levels, h.coarse_inv, 1, 1, 1, fused=fused)(b)
            assert np.abs(z - x).max() <= 1e-10 * np.abs(x).max(), fused
    # not applied where it does not pay: a mesh-like graph keeps more than 30 % of its unknowns outside any independent set
    m = 30
    T1 = sp.diags([-1.0, 2.0, -1.0], [-1, 0, 1], shape=(m, m))
    Lap = (sp.kron(sp.identity(m), T1) + sp.kron(T1, sp.identity(m)) + 1e-3 * sp.identity(m * m)).tocsr()
    assert amg.try_elimination_level(Lap, Lap.diagonal(), 1.0 / Lap.diagonal(), np.ones(m * m, bool), np.zeros(m * m, bool), 2.0, 2500, 1) is None


def test_rise_and_decay_stimulus_expression():
    """``HodgkinHuxley._add_stimulus(step=False)``: the gradually rising stimulus exp(-t/tau_decay) - exp(-t/tau_rise) (reference
    KNPEMIx_ionic_model.py:553-555, parameters mixed_dim_problem.py:299-304).  The reference's own wiring always passes step=True
    (KNPEMIx_problem.py:538-542), so no configuration reaches this branch; the expression itself -- compiled to the membrane
    bytecode and interpreted -- is checked against the formula, with the area scaling and a region mask."""
    from CGx.KNPEMI.KNPEMIx_ionic_model import HodgkinHuxley
    from CGx.KNPEMI.KNPEMIx_problem import ProblemKNPEMI
    cfg = ci_config(N=8, steps=2)
    cfg["stimulus"].update({"tau_syn_rise": 2e-4, "tau_syn_decay": 1e-3})
    cfg["stimulus_region"] = {"direction": "x", "range": [0.2, 0.6]}
    p = ProblemKNPEMI(cfg)
    assert p.tau_syn_rise == 2e-4 and p.tau_syn_decay == 1e-3
    hh = HodgkinHuxley(p)
    p.set_initial_conditions()
    p.init_ionic_models([hh])
    p.setup_variational_form()                      # defines the Nernst potentials ion["E"]
    t = 7.3e-4
    hh.t_mod.value = t
    expr = hh._add_stimulus(0, step=False, range=p.stimulus_region_range, dir=p.stimulus_region_direction)
    o = K.make_square(8, models=[K.Model("hh", (4,))])
    o.stimulus_region = (0, 0.2e-6, 0.6e-6)
    area = float((o.fmeas * o._facet_mask_integral()).sum())
    assert abs(p.stimulus_area - area) <= 1e-12 * area
    rng = np.random.default_rng(2)
    nq = 50
    phim = -0.07 + 0.01 * rng.random(nq)
    na_i, na_e = 12.0 + rng.random(nq), 140.0 + rng.random(nq)
    x = [1e-6 * rng.random(nq), 1e-6 * rng.random(nq)]
    roles = {id(p.phi_m_prev): ("PHIM", 0)}
    for j in range(3):
        roles[id(p.wh[0][j])] = ("KI", j)
        roles[id(p.wh[1][j])] = ("KE", j)
    spec = fem.compile_program([expr, 0.0 * expr, 0.0 * expr], roles)
    out = fem.interpret_program(spec, [na_i, 0 * na_i + 130, 0 * na_i + 5], [na_e, 0 * na_e + 4, 0 * na_e + 125], phim, [], x)
    E_na = p.psi.value * np.log(na_e / na_i)
    mask = ((x[0] > 0.2e-6) & (x[0] < 0.6e-6)) * 1.0
    want = mask * 1e-9 * (np.exp(-t / 1e-3) - np.exp(-t / 2e-4)) * (phim - E_na) / area
    assert np.max(np.abs(out[0] - want)) <= 1e-13 * np.max(np.abs(want)) and np.abs(want).max() > 0


def test_multilevel_kway_partition_beats_coordinate_bisection():
    """The native counterpart of the graph partitioner the reference inherits from DOLFINx (mixed_dim_problem.py:21,649,666):
    multilevel k-way partition of the nodal graph, vertex weight = unknowns per vertex (membrane vertices carry two nodes), heavier
    edges inside cells.  On the tissue lattice and on an irregular (Delaunay) mesh it is balanced within its 3 % band, deterministic,
    covers every part, and cuts clearly less edge weight than the weighted coordinate bisection it replaces for meshes read from files."""
    from scipy.spatial import Delaunay
    from cgx_hip import partition as PT
    cases = []
    # (9 cells per direction: no bisection plane falls on an extracellular sheet, as it does by luck for 6 or 8 cells)
    coords, cells, tags, _, _ = meshmod.load_mesh("tissue3d_37_9_w1.xdmf", "tissue3d_37_9_w1.xdmf", 1.0)
    cases.append(("tissue", coords, cells, tags != 1, 0.75))
    rng = np.random.default_rng(0)
    pts = rng.random((20000, 2))
    pts = pts[~((pts[:, 0] > 0.5) & (pts[:, 1] > 0.5))]                       # L-shaped domain
    tri = Delaunay(pts).simplices
    cen = pts[tri].mean(axis=1)
    tri = tri[~((cen[:, 0] > 0.5) & (cen[:, 1] > 0.5))]
    cen = pts[tri].mean(axis=1)
    cases.append(("irregular", pts, tri, np.hypot(cen[:, 0] - 0.25, cen[:, 1] - 0.25) < 0.15, 0.95))
    for name, X, T, intra, factor in cases:
        vw = PT.mesh_vertex_weights(len(X), T, intra)
        G = PT.nodal_graph(len(X), T, np.where(intra, 4.0, 1.0))
        assert abs(G - G.T).max() == 0
        for k in (5, 8):
            own = PT.partition_mesh_vertices(X, T, k, intra)
            assert np.array_equal(own, PT.partition_mesh_vertices(X, T, k, intra))          # deterministic
            pw = np.bincount(own, weights=vw, minlength=k)
            assert own.min() == 0 and own.max() == k - 1 and pw.min() > 0
            assert pw.max() <= 1.035 * pw.mean(), (name, k, pw)
            rcb = vertex_partition(X, k, weights=vw)
            assert PT.edge_cut(G, own) <= factor * PT.edge_cut(G, rcb), (name, k, PT.edge_cut(G, own), PT.edge_cut(G, rcb))
    # partition_mesh(method="kway") yields consistent local meshes: every vertex owned exactly once, ghost owners right
    X, T, intra = cases[0][1], cases[0][2], cases[0][3]
    tg = np.where(intra, 2, 1).astype(np.int32)
    g, gt, _ = meshmod.gamma_integration_entities(T, tg, (2,), (1,))
    seen = np.zeros(len(X), dtype=int)
    for r in range(4):
        lm = partition_mesh(X, T, tg, g, gt, 4, r, intra_tags=(2,), method="kway")
        seen[lm.l2g[:lm.n_vertices_owned]] += 1
    assert (seen == 1).all()


def test_morton_reordering_keeps_the_mesh():
    """cgx_hip/parallel.py reorder_local_mesh: the renumbered local mesh is the same mesh -- same cells as sets of global vertices, same
    coordinates per global vertex, same membrane facets with matching local facet indices on both sides, tags carried along, owned
    entities still in front -- and the Morton keys really follow a Z curve (a 2x2x2 block of lattice points is contiguous)."""
    import numpy as np
    from cgx_hip.parallel import morton_keys, reorder_local_mesh, stacked_cubes_local_mesh
    lm = stacked_cubes_local_mesh(8, 1, 0, scale=1e-6)
    m2 = reorder_local_mesh(lm, "morton")
    assert reorder_local_mesh(lm, "native") is lm
    assert m2.n_vertices_owned == lm.n_vertices_owned and m2.n_cells_owned == lm.n_cells_owned
    assert np.array_equal(np.sort(m2.l2g), np.sort(lm.l2g)) and not np.array_equal(m2.l2g, lm.l2g)
    assert np.allclose(lm.coords[np.argsort(lm.l2g)], m2.coords[np.argsort(m2.l2g)])

    def cell_table(m):
        return {tuple(sorted(m.l2g[c])): int(t) for c, t in zip(m.cells, m.cell_tags)}
    assert cell_table(lm) == cell_table(m2)

    def facets(m):
        out = {}
        for (cp, lfp, cm, lfm), t in zip(m.gamma, m.gamma_tags):
            vp = sorted(m.l2g[v] for i, v in enumerate(m.cells[cp]) if i != lfp)
            vm = sorted(m.l2g[v] for i, v in enumerate(m.cells[cm]) if i != lfm)
            assert vp == vm
            out[tuple(vp)] = (int(t), int(m.cell_tags[cp]), int(m.cell_tags[cm]))
        return out
    assert facets(lm) == facets(m2) and len(lm.gamma) > 0
    g = np.array([[i, j, k] for k in range(4) for j in range(4) for i in range(4)], dtype=float)
    order = np.argsort(morton_keys(g), kind="stable")
    first8 = {tuple(g[i].astype(int)) for i in order[:8]}
    assert first8 == {(i, j, k) for i in (0, 1) for j in (0, 1) for k in (0, 1)}
    with __import__("pytest").raises(ValueError):
        reorder_local_mesh(lm, "hilbert")
    # partitioned meshes: ghosts stay behind the owned vertices and keep their owners
    from cgx_hip import mesh as M
    from cgx_hip.parallel import partition_mesh
    coords, cells, tags, _, _ = M.load_mesh("cube6.xdmf", "cube6.xdmf", 1e-6)
    intra = tuple(int(t) for t in np.unique(tags) if t != 1)
    gam, gt, _ = M.gamma_integration_entities(cells, tags, intra, (1,), "intra")
    parts = [reorder_local_mesh(partition_mesh(coords, cells, tags, gam, gt, 3, r, intra_tags=intra), "morton") for r in range(3)]
    owned = [set(p.l2g[:p.n_vertices_owned].tolist()) for p in parts]
    assert sum(len(o) for o in owned) == coords.shape[0]
    for p in parts:
        assert len(set(p.l2g.tolist())) == len(p.l2g)
        assert all(int(g) in owned[int(o)] for g, o in zip(p.l2g[p.n_vertices_owned:], p.ghost_owner))

