"""Run artefacts of the reference's solver that need no I/O library: probe-point evaluation, membrane traces, ``.npy``
exports and nodal-field checkpoints.

Mirrors reference
  src/CGx/utils/mixed_dim_problem.py:277-287    ``point_evaluation`` config key (points scaled by mesh_conversion_factor)
  src/CGx/utils/mixed_dim_problem.py:496-632    membrane measurement vertex closest to the mesh centre (``png_dof``, owner rank)
  src/CGx/utils/mixed_dim_problem.py:744-804    default ``gamma_points`` (measurement vertex, or a vertex inside the stimulus region)
  src/CGx/KNPEMI/KNPEMIx_solver.py:551-643      init_png_savefile / save_png / init_data / save_data
  src/CGx/KNPEMI/KNPEMIx_solver.py:799-821      checkpoints of the 2(N+1) solution functions every ``save_interval`` steps
  src/CGx/KNPEMI/KNPEMIx_solver.py:833-866      export_data: file names of the ``.npy`` artefacts
The reference evaluates with scifem.evaluate_function (P1 interpolation in the cell containing the point) and checkpoints with
adios4dolfinx; here the interpolation weights are found once on the host and each evaluation is one tiny device gather, and a
checkpoint is an ``.npz`` of the nodal arrays per rank (with the local-to-global vertex map).
"""
from __future__ import annotations

import os

import numpy as np
import torch


def _barycentric(coords, cells, pts, tol=1e-9):
    """For every point the first cell (lowest index) that contains it: returns (cell index or -1, weights (n_pts, d+1))."""
    d = coords.shape[1]
    X0 = coords[cells[:, 0]]                                            # (nc, d)
    T = np.stack([coords[cells[:, a + 1]] - X0 for a in range(d)], axis=2)   # (nc, d, d): columns = edge vectors
    lo = coords[cells].min(axis=1)
    hi = coords[cells].max(axis=1)
    scale = float((hi - lo).max()) if len(cells) else 1.0
    out_c = np.full(len(pts), -1, dtype=np.int64)
    out_w = np.zeros((len(pts), d + 1))
    for k, p in enumerate(pts):
        cand = np.nonzero(np.all((lo <= p + tol * scale) & (hi >= p - tol * scale), axis=1))[0]
        if cand.size == 0:
            continue
        lam = np.linalg.solve(T[cand], (p - X0[cand])[:, :, None])[:, :, 0]     # (n_cand, d)
        w = np.concatenate([1.0 - lam.sum(axis=1, keepdims=True), lam], axis=1)
        ok = np.nonzero(np.all(w >= -tol, axis=1))[0]
        if ok.size:
            j = ok[0]
            out_c[k] = cand[j]
            out_w[k] = np.clip(w[j], 0.0, 1.0)
            out_w[k] /= out_w[k].sum()
    return out_c, out_w


class PointEvaluator:
    """P1 evaluation of nodal fields at fixed points.  ``side``: 0 search intracellular cells only, 1 extracellular only,
    None any cell (membrane points: the weight of the vertex opposite the facet is zero either way)."""

    def __init__(self, problem, points, side=None):
        p = problem
        lm = p.local_mesh
        pts = np.atleast_2d(np.asarray(points, dtype=np.float64))[:, :lm.coords.shape[1]]
        sel = np.arange(lm.n_cells_owned)
        if side is not None:
            sel = sel[p.cell_side[:lm.n_cells_owned] == side]
        c, w = _barycentric(lm.coords, lm.cells[sel], pts)
        found = c >= 0
        # the lowest rank that found a point evaluates it
        owners = p.comm.all_gather_object(found.tolist())
        owner = np.full(len(pts), -1, dtype=np.int64)
        for r in range(p.comm.size - 1, -1, -1):
            owner[np.asarray(owners[r], dtype=bool)] = r
        if (owner < 0).any():
            bad = pts[owner < 0] / getattr(p, "mesh_conversion_factor", 1.0)
            raise RuntimeError(f"point_evaluation: no {'cell' if side is None else ('intracellular', 'extracellular')[side] + ' cell'} "
                               f"contains the point(s) {bad.tolist()} (mesh units)")
        self.comm = p.comm
        self.n_points = len(pts)
        self.mine = np.nonzero(owner == p.comm.rank)[0]
        dev = p.mesh.device
        verts = lm.cells[sel[c[self.mine]]] if self.mine.size else np.zeros((0, lm.cells.shape[1]), dtype=np.int64)
        self.verts = torch.as_tensor(np.ascontiguousarray(verts, dtype=np.int64), device=dev)
        self.weights = torch.as_tensor(np.ascontiguousarray(w[self.mine]), dtype=torch.float64, device=dev)

    def __call__(self, functions):
        """values[f, point] of the given Functions (one device gather + one small copy per call)."""
        out = np.zeros((len(functions), self.n_points))
        if self.mine.size:
            stack = torch.stack([(f.x.array[self.verts] * self.weights).sum(dim=1) for f in functions])
            out[:, self.mine] = stack.cpu().numpy()
        if self.comm.size > 1:
            parts = self.comm.all_gather_object(out)
            out = np.sum(parts, axis=0)
        return out


def find_membrane_measurement_vertex(problem):
    """The membrane vertex (of the facets tagged ``membrane_data_tag``) closest to the centre of the mesh's bounding box:
    sets ``png_point``, ``png_dof`` (local vertex index on the owner), ``owner_rank_membrane_vertex`` and, when the config
    gave none, ``gamma_points`` (reference mixed_dim_problem.py:496-632, 744-804)."""
    p = problem
    lm = p.local_mesh
    mm = p.get_min_and_max_coordinates()
    d = lm.coords.shape[1]
    centre = np.array([(mm[2 * a] + mm[2 * a + 1]) / 2 for a in range(d)])
    tags = p.gamma_tags if p.MMS_test else (p.membrane_data_tag,)
    sel = np.isin(p.gamma_facet_tags, tags)
    gv = np.unique(p._fv[sel]) if sel.any() else np.zeros(0, dtype=np.int64)
    gv = gv[gv < lm.n_vertices_owned]
    if gv.size:
        dist = ((lm.coords[gv] - centre) ** 2).sum(axis=1)
        k = int(np.argmin(dist))
        mine = (float(dist[k]), p.comm.rank, int(gv[k]), lm.coords[gv[k]].tolist())
    else:
        mine = (np.inf, p.comm.rank, -1, None)
    best = min(p.comm.all_gather_object(mine), key=lambda t: (t[0], t[1]))
    if best[2] < 0:
        raise RuntimeError(f"no membrane facet carries membrane_data_tag {tags}")
    p.owner_rank_membrane_vertex = best[1]
    p.png_dof = best[2]
    p.png_point = np.array([best[3]])
    p.print("Phi m measurement point: ", p.png_point[0])
    if p.gamma_points is None:
        if not p.MMS_test and p.stimulus_region:
            x = lm.coords[gv] if gv.size else np.zeros((0, d))
            mask = np.ones(len(x), dtype=bool)
            if p.multiple_stimulus_directions:
                for i, ax in enumerate(p.stimulus_region_directions):
                    mask &= (x[:, ax] > p.stimulus_region_range[i][0]) & (x[:, ax] < p.stimulus_region_range[i][1])
            else:
                ax, rng = p.stimulus_region_direction, p.stimulus_region_range
                mask &= (x[:, ax] > rng[0]) & (x[:, ax] < rng[1])
            local = x[mask][0].tolist() if mask.any() else None
            pts = [q for q in p.comm.all_gather_object(local) if q is not None]
            p.gamma_points = np.array([pts[0]]) if pts else p.png_point
        else:
            p.gamma_points = p.png_point


class RunOutput:
    """Everything ``SolverKNPEMI`` records besides the solve itself; one instance per solver."""

    def __init__(self, solver):
        self.s = solver
        p = solver.problem
        self.p = p
        self.prefix = p.output_dir
        self.traces = bool(solver.save_pngs or solver.save_dat)
        self.v_t, self.n_t, self.m_t, self.h_t = [], [], [], []
        self.ics_eval = self.ecs_eval = self.gamma_eval = None
        if self.traces or p.point_evaluation:
            os.makedirs(self.prefix, exist_ok=True)
            if not hasattr(p, "png_dof"):
                find_membrane_measurement_vertex(p)
        if p.point_evaluation:                                         # KNPEMIx_solver.py:612-626
            n = solver.time_steps + 1
            self.ics_eval = PointEvaluator(p, p.ics_points, side=0)
            self.ecs_eval = PointEvaluator(p, p.ecs_points, side=1)
            self.gamma_eval = PointEvaluator(p, p.gamma_points, side=None)
            solver.ics_point_values = np.zeros((n, p.num_variables, len(p.ics_points)))
            solver.ecs_point_values = np.zeros((n, p.num_variables, len(p.ecs_points)))
            solver.gamma_point_values = np.zeros((n, len(p.gamma_points)))
        if solver.save_cpoints:
            os.makedirs(os.path.join(self.prefix, "checkpoints"), exist_ok=True)
        self.xdmf = None
        if getattr(solver, "save_xdmfs", False):
            os.makedirs(self.prefix, exist_ok=True)
            self.init_xdmf_savefile()

    # ---- KNPEMIx_solver.py:551-610
    def _trace_point(self):
        p = self.p
        if p.comm.rank != p.owner_rank_membrane_vertex:
            return
        d = p.png_dof
        self.v_t.append(1000.0 * float(p.phi_m_prev.x.array[d]))       # mV
        if hasattr(p, "n"):
            self.n_t.append(float(p.n.x.array[d]))
            self.m_t.append(float(p.m.x.array[d]))
            self.h_t.append(float(p.h.x.array[d]))

    def record(self, i):
        """state after timestep i (i = 0: initial data)"""
        s, p = self.s, self.p
        if self.traces:
            self._trace_point()
        if p.point_evaluation:                                         # KNPEMIx_solver.py:628-643
            s.ics_point_values[i] = self.ics_eval(p.wh[0])
            s.ecs_point_values[i] = self.ecs_eval(p.wh[1])
            s.gamma_point_values[i] = self.gamma_eval([p.phi_m_prev])[0]
        if s.save_cpoints and (i % s.save_interval == 0):
            self.checkpoint(i)
        if self.xdmf is not None and i > 0 and (i % s.save_interval == 0):       # reference :471
            self.save_xdmf()

    # ---- KNPEMIx_solver.py:766-797, 824-831: subdomains.xdmf (mesh + cell tags) and solution.xdmf (time series of the 2(N+1)
    # nodal functions), heavy data in .h5 files next to them (cgx_hip/hdf5_write.py); one pair of files per rank
    def _xdmf_names(self, stem):
        p = self.p
        sfx = f"_rank{p.comm.rank}" if p.comm.size > 1 else ""
        return os.path.join(self.prefix, stem + sfx + ".xdmf"), os.path.join(self.prefix, stem + sfx + ".h5")

    def _mesh_items(self, h5name, n_cells, n_pts, dim):
        ttype = "Triangle" if dim == 2 else "Tetrahedron"
        return [f'      <Topology TopologyType="{ttype}" NumberOfElements="{n_cells}" NodesPerElement="{dim + 1}">',
                f'        <DataItem Dimensions="{n_cells} {dim + 1}" NumberType="Int" Precision="8" Format="HDF">{h5name}:/Mesh/mesh/topology</DataItem>',
                f'      </Topology>',
                f'      <Geometry GeometryType="{"XY" if dim == 2 else "XYZ"}">',
                f'        <DataItem Dimensions="{n_pts} {dim}" NumberType="Float" Precision="8" Format="HDF">{h5name}:/Mesh/mesh/geometry</DataItem>',
                f'      </Geometry>']

    def _tag_grid(self, h5name, n_cells, n_pts, dim):
        return ['    <Grid Name="ct" GridType="Uniform">'] + self._mesh_items(h5name, n_cells, n_pts, dim)[:3] + [
            f'      <Geometry GeometryType="{"XY" if dim == 2 else "XYZ"}">',
            f'        <DataItem Dimensions="{n_pts} {dim}" NumberType="Float" Precision="8" Format="HDF">{h5name}:/Mesh/mesh/geometry</DataItem>',
            f'      </Geometry>',
            f'      <Attribute Name="ct" AttributeType="Scalar" Center="Cell">',
            f'        <DataItem Dimensions="{n_cells} 1" NumberType="Int" Precision="4" Format="HDF">{h5name}:/MeshTags/ct/Values</DataItem>',
            f'      </Attribute>',
            f'    </Grid>']

    @staticmethod
    def _xdmf_text(body):
        return "\n".join(['<?xml version="1.0"?>', '<!DOCTYPE Xdmf SYSTEM "Xdmf.dtd" []>', '<Xdmf Version="3.0">', '  <Domain>'] + body + ['  </Domain>', '</Xdmf>', ''])

    def init_xdmf_savefile(self):
        from .hdf5_write import Hdf5Writer
        p = self.p
        lm = p.local_mesh
        coords = np.asarray(lm.coords, dtype=np.float64)
        cells = np.asarray(lm.cells, dtype=np.int64)
        tags = np.asarray(lm.cell_tags, dtype=np.int32).reshape(-1, 1)
        dim = cells.shape[1] - 1
        nc, npt = len(cells), len(coords)
        for stem in ("subdomains", "solution"):
            xname, hname = self._xdmf_names(stem)
            w = Hdf5Writer(hname)
            w.write("/Mesh/mesh/geometry", coords)
            w.write("/Mesh/mesh/topology", cells)
            w.write("/MeshTags/ct/topology", cells)
            w.write("/MeshTags/ct/Values", tags)
            base = os.path.basename(hname)
            head = ['    <Grid Name="mesh" GridType="Uniform">'] + self._mesh_items(base, nc, npt, dim) + ['    </Grid>'] + self._tag_grid(base, nc, npt, dim)
            if stem == "subdomains":
                w.close()
                with open(xname, "w") as f:
                    f.write(self._xdmf_text(head))
            else:
                self.xdmf = {"writer": w, "xname": xname, "h5": base, "head": head, "steps": [], "dims": (nc, npt, dim)}
        self.save_xdmf()                               # initial state, reference :787-790

    def save_xdmf(self):
        """the 2(N+1) solution functions at the current time (reference :792-797), appended to solution.h5"""
        if self.xdmf is None:
            return
        p = self.p
        X = self.xdmf
        k = len(X["steps"])
        names = []
        for idx in range(p.num_variables):
            for side in (0, 1):
                f = p.wh[side][idx]
                X["writer"].write(f"/Function/{f.name}/{k}", np.asarray(f.numpy(), dtype=np.float64).reshape(-1, 1))
                names.append(f.name)
        X["steps"].append((float(p.t.value), names))
        self._write_solution_xml()

    _XML_TAIL = ['    </Grid>', '  </Domain>', '</Xdmf>', '']

    def _write_solution_xml(self):
        """solution.xdmf is always a complete document: the text of the steps not yet written is put where the closing tags of
        the previous version began, followed by the closing tags again -- O(steps) text in total (regenerating the
        whole file at every save was O(steps^2))."""
        X = self.xdmf
        nc, npt, dim = X["dims"]
        if "xml_pos" not in X:
            head = self._xdmf_text(list(X["head"]) + ['    <Grid Name="solution" GridType="Collection" CollectionType="Temporal">'])
            head = head[:head.index("  </Domain>")]
            with open(X["xname"], "w") as f:
                f.write(head)
                X["xml_pos"] = f.tell()
                f.write("\n".join(self._XML_TAIL))
            X["xml_steps"] = 0
        body = []
        for k in range(X["xml_steps"], len(X["steps"])):
            t, names = X["steps"][k]
            body += [f'      <Grid Name="step_{k}" GridType="Uniform">', f'        <Time Value="{t!r}" />']
            body += ["  " + ln for ln in self._mesh_items(X["h5"], nc, npt, dim)]
            for nm in names:
                body += [f'        <Attribute Name="{nm}" AttributeType="Scalar" Center="Node">',
                         f'          <DataItem Dimensions="{npt} 1" NumberType="Float" Precision="8" Format="HDF">{X["h5"]}:/Function/{nm}/{k}</DataItem>',
                         f'        </Attribute>']
            body += ['      </Grid>']
        if not body:
            return
        with open(X["xname"], "r+") as f:
            f.seek(X["xml_pos"])
            f.write("\n".join(body) + "\n")
            X["xml_pos"] = f.tell()
            f.write("\n".join(self._XML_TAIL))
            f.truncate()
        X["xml_steps"] = len(X["steps"])

    def close_xdmf(self):
        if self.xdmf is not None:
            self.xdmf["writer"].close()                # the .h5 file becomes readable here (metadata + superblock)
            self._write_solution_xml()
            self.p.print("\nXDMF output saved in ", self.prefix)
            self.xdmf = None

    def checkpoint(self, i):
        """nodal values of the 2(N+1) solution functions (+ phi_m and the gating variables) of this rank's owned vertices"""
        p = self.p
        lm = p.local_mesh
        nvo = lm.n_vertices_owned
        data = {"t": float(p.t.value), "step": i, "l2g": lm.l2g[:nvo], "coords": lm.coords[:nvo]}
        for f in p.u_out_i + p.u_out_e:
            data[f.name] = f.numpy()[:nvo]
        data["phi_m"] = p.phi_m_prev.numpy()[:nvo]
        for nm in ("n", "m", "h"):
            if hasattr(p, nm):
                data[nm] = getattr(p, nm).numpy()[:nvo]
        np.savez(os.path.join(self.prefix, "checkpoints", f"step_{i:06d}_rank{p.comm.rank}.npz"), **data)

    # ---- KNPEMIx_solver.py:833-866 (same file names)
    def export(self):
        s, p = self.s, self.p
        out = p.output_dir
        if self.traces and p.comm.rank == p.owner_rank_membrane_vertex:
            np.save(out + "phi_m.npy", np.array(self.v_t))
            if hasattr(p, "n"):
                np.save(out + "n.npy", np.array(self.n_t))
                np.save(out + "m.npy", np.array(self.m_t))
                np.save(out + "h.npy", np.array(self.h_t))
        if p.comm.rank == 0:
            if p.point_evaluation:
                np.save(out + "gamma_point_values.npy", s.gamma_point_values)
                np.save(out + "ics_point_values.npy", s.ics_point_values)
                np.save(out + "ecs_point_values.npy", s.ecs_point_values)
            np.save(out + "assembly_time.npy", np.array(s.assembly_time))
            np.save(out + "solve_time.npy", np.array(s.solve_time))
            if not s.direct_solver:
                np.save(out + "iterations.npy", np.array(s.iterations))

    def figures(self):
        """PNG plots of the traces (KNPEMIx_solver.py:645-764) when matplotlib is installed; the data are exported either way."""
        try:
            import matplotlib
            matplotlib.use("Agg")
            import matplotlib.pyplot as plt
        except Exception:      # noqa: BLE001
            self.p.print("matplotlib is not installed: PNG figures skipped, traces are in the .npy files")
            return
        s, p = self.s, self.p
        times = np.linspace(0, 1000 * s.time_steps * float(p.dt.value), s.time_steps + 1)
        if p.comm.rank == p.owner_rank_membrane_vertex and self.v_t:
            fig, ax = plt.subplots()
            ax.plot(times[:len(self.v_t)], np.array(self.v_t))
            ax.set_xlabel("Time [ms]"); ax.set_ylabel("Membrane potential [mV]")
            fig.savefig(self.prefix + "v.png")
            if self.n_t:
                fig, ax = plt.subplots()
                for arr, lab in ((self.n_t, "n"), (self.m_t, "m"), (self.h_t, "h")):
                    ax.plot(times[:len(arr)], np.array(arr), label=lab)
                ax.set_xlabel("Time [ms]"); ax.legend()
                fig.savefig(self.prefix + "gating.png")
        if p.comm.rank == 0:
            fig, ax = plt.subplots()
            ax.plot(s.assembly_time, label="assembly"); ax.plot(s.solve_time, label="solve")
            ax.set_xlabel("Timestep"); ax.set_ylabel("Time [s]"); ax.legend()
            fig.savefig(self.prefix + "timings.png")
            if not s.direct_solver:
                fig, ax = plt.subplots()
                ax.plot(s.iterations); ax.set_xlabel("Timestep"); ax.set_ylabel("Number of iterations")
                fig.savefig(self.prefix + "iterations.png")
