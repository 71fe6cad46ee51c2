"""``MixedDimensionalProblem`` / ``ProblemKNPEMI``: the reference's problem-definition surface on top
of the MI355X-native backend.

Mirrors (names, constructor, attributes, call order) reference
  src/CGx/utils/mixed_dim_problem.py   (YAML schema :86-374, tags :376-433, ionic models :435-465,
                                        domain / '+'=intra ordering :634-733)
  src/CGx/KNPEMI/KNPEMIx_problem.py    (spaces/restrictions :28-94, ICs :220-452, forms :454-655,
                                        preconditioner :657-744, constants :909-981)
What differs by design: there is no UFL/FFCx/DOLFINx.  ``setup_variational_form`` compiles the
membrane currents of the registered mechanisms to bytecode programs for the HIP membrane kernel;
the block forms a, L, P themselves are hard-wired in the HIP assembly kernels
(knp-emi-cgx_amd/csrc/knp_kernels.hip), which restate KNPEMIx_problem.py:586-642 and :717-738.
"""
from __future__ import annotations

import collections.abc
import ctypes as C
import math
import os
import pathlib
import time
from abc import ABC, abstractmethod

import numpy as np
import torch
import yaml

from . import _lib, fem
from . import mesh as meshmod
from .fem import Constant, Function, FunctionSpace
from .parallel import Comm, LocalMesh, partition_mesh


def range_constructor(loader, node):
    """``!range [a, b]`` YAML tag (reference src/CGx/utils/misc.py:33-37)."""
    args = loader.construct_sequence(node)
    return list(range(*args))


def flatten_list(input_list):
    return [item for sublist in input_list for item in (sublist if isinstance(sublist, tuple) else [sublist])]


class _Measure:
    """Placeholder for ufl.Measure so that ``problem.dx(tags)`` / ``problem.dS(tags)`` stay callable;
    integration itself happens in the HIP kernels."""

    def __init__(self, kind, degree=10):
        self.kind, self.degree, self.tags = kind, degree, None

    def __call__(self, tags=None):
        m = _Measure(self.kind, self.degree)
        m.tags = tags
        return m


class MixedDimensionalProblem(ABC):

    def __init__(self, config_file: str, comm: Comm | None = None, local_mesh: LocalMesh | None = None):
        tic = time.perf_counter()
        self.comm = comm if comm is not None else Comm()
        self._local_mesh_override = local_mesh
        self.read_config_file(config_file=config_file)
        self.print("Read input data from " + (str(config_file) if not isinstance(config_file, dict) else "<dict>"))
        self.setup_domain()
        self.t = Constant(self.mesh, 0.0)
        self.dt = Constant(self.mesh, self.dt)
        self.setup_constants()
        self.setup_spaces()
        self.init()
        self.setup_boundary_conditions()
        if self.source_terms == "ion_injection":
            self.setup_source_terms()
        self.ionic_models = []
        self.backend = None
        self.print(f"Problem setup in {time.perf_counter() - tic:0.4f} seconds.\n")

    def print(self, *a, **k):
        if self.comm.rank == 0 and not getattr(self, "quiet", False):
            print(*a, **k, flush=True)

    @abstractmethod
    def init(self): ...
    @abstractmethod
    def setup_spaces(self): ...
    @abstractmethod
    def setup_boundary_conditions(self): ...
    @abstractmethod
    def setup_source_terms(self): ...
    @abstractmethod
    def setup_constants(self): ...

    # ---------------------------------------------------------------- config (schema of :86-374)
    def read_config_file(self, config_file):
        yaml.add_constructor("!range", range_constructor, Loader=yaml.FullLoader)
        if isinstance(config_file, dict):
            config = config_file
        else:
            with open(config_file, "r") as file:
                config = yaml.load(file, Loader=yaml.FullLoader)
        self.config = config
        self.quiet = bool(config.get("quiet", False))
        if "solver" in config:
            self.solver_config: dict = config["solver"]
        else:
            raise RuntimeError("Provide solver configuration in input file.")
        input_dir = config.get("input_dir", "./")
        if "output_dir" in config:
            self.output_dir = config["output_dir"]
        else:
            self.output_dir = "./output/"
        if any(self.solver_config.get("output", {}).get(k, False) for k in ("save_xdmf", "save_pngs", "save_cpoints", "save_dat", "save_mat")):
            pathlib.Path(self.output_dir).mkdir(parents=True, exist_ok=True)
        if "cell_tag_file" in config and "facet_tag_file" in config:
            self.input_files = {"mesh_file": input_dir + config["cell_tag_file"],
                                "facet_file": input_dir + config["facet_tag_file"]}
        else:
            raise RuntimeError("Provide cell_tag_file and facet_tag_file fields in input file.")
        if "dt" in config:
            self.dt = float(config["dt"])
        else:
            raise RuntimeError("Provide dt (timestep size) field in input file.")
        if "time_steps" in config:
            self.time_steps = int(config["time_steps"])
        elif "T" in config:
            self.time_steps = int(float(config["T"]) / float(config["dt"]))
        else:
            raise RuntimeError("Provide final time T or time_steps field in input file.")

        tags = {}
        if "ics_tags" in config:
            tags["intra"] = config["ics_tags"]
        else:
            raise RuntimeError("Provide ics_tags (intracellular space tags) field in input file.")
        if "ecs_tags" in config: tags["extra"] = config["ecs_tags"]
        if "boundary_tags" in config: tags["boundary"] = config["boundary_tags"]
        if "membrane_tags" in config: tags["membrane"] = config["membrane_tags"]
        if "stimulus_tags" in config:
            self.stimulus_tags = config["stimulus_tags"]
        else:
            self.stimulus_tags = tags.get("membrane", tags["intra"])
        if "glia_tags" in config:
            tags["glia"] = config["glia_tags"]
            tags["neuron"] = [t for t in _aslist(tags["intra"]) if t not in tags["glia"]]
        else:
            tags["neuron"] = tags["intra"]
        self.parse_tags(tags=tags)

        if "physical_constants" in config:
            consts = config["physical_constants"]
            self.T_value = float(consts.get("T", 1.0))
            self.R_value = float(consts.get("R", 1.0))
            self.F_value = float(consts.get("F", 1.0))
            self.psi_value = self.R_value * self.T_value / self.F_value
        else:
            self.T_value = self.R_value = self.F_value = self.psi_value = 1.0
        self.C_M_value = float(config.get("C_M", 1.0))
        if "mesh_conversion_factor" in config:
            self.mesh_conversion_factor = float(config["mesh_conversion_factor"])
        if "fem_order" in config:
            self.fem_order = int(config["fem_order"])
            if self.fem_order != 1:
                raise RuntimeError("The MI355X-native path implements P1 elements only (fem_order: 1).")
        if "dirichlet_bcs" in config:
            self.dirichlet_bcs = bool(config["dirichlet_bcs"])
        if "MMS_test" in config:
            self.MMS_test = True
            self.dirichlet_bcs = True
            try:
                self.N_mesh = int(config["MMS_test"]["N_mesh"])
            except Exception:
                raise RuntimeError('For MMS test, provide number of mesh cells "N_mesh" in input file.')
            try:
                self.dim = int(config["MMS_test"]["dim"])
            except Exception:
                raise RuntimeError('For MMS test, provide dimension "dim" in input file.')
        self.source_terms = config.get("source_terms", None)
        if self.source_terms not in (None, "ion_injection"):
            raise RuntimeError(f"Unknown source_terms '{self.source_terms}' (the reference knows 'ion_injection').")
        if "point_evaluation" in config:                     # mixed_dim_problem.py:277-287
            pe = config["point_evaluation"]
            self.point_evaluation = True
            self.ics_points = np.array(pe["ics_points"], dtype=np.float64) * self.mesh_conversion_factor
            self.ecs_points = np.array(pe["ecs_points"], dtype=np.float64) * self.mesh_conversion_factor
            self.gamma_points = np.array(pe["gamma_points"], dtype=np.float64) * self.mesh_conversion_factor if "gamma_points" in pe else None
        else:
            self.point_evaluation = False
            self.gamma_points = None

        if "stimulus" in config:
            try:
                g_dict = config["stimulus"]["conductance"]
                self.g_syn_bar_val = float(g_dict["g_syn_bar"])
                self.a_syn_val = float(config["stimulus"]["a_syn"])
                self.T_stim_val = float(config["stimulus"]["T_stim"])
            except Exception:
                raise RuntimeError("For stimulus, provide g_syn_bar, a_syn and T_stim in input file.")
            if "tau_syn_rise" in config["stimulus"] or "tau_syn_decay" in config["stimulus"]:
                try:
                    self.tau_syn_rise = float(config["stimulus"]["tau_syn_rise"])
                    self.tau_syn_decay = float(config["stimulus"]["tau_syn_decay"])
                except Exception:
                    raise RuntimeError("For rise and decay stimulus, provide tau_syn_rise and tau_syn_decay in input file.")
            if "scale" in config["stimulus"]:
                self.scale_stimulus = bool(config["stimulus"]["scale"])
            else:
                raise RuntimeError("Provide whether to scale stimulus strength by surface area in stimulus configuration in input file.")
            self.g_Na_bar_val = float(g_dict.get("g_Na_bar", 1200.0))
            self.g_K_bar_val = float(g_dict.get("g_K_bar", 360.0))
            self.g_Na_leak_val = float(g_dict.get("g_Na_leak", 0.3))
            self.g_Na_leak_g_val = float(g_dict.get("g_Na_leak_g", 1.0))
            self.g_K_leak_val = float(g_dict.get("g_K_leak", 0.1))
            self.g_K_leak_g_val = float(g_dict.get("g_K_leak_g", 16.96))
            self.g_Cl_leak_val = float(g_dict.get("g_Cl_leak", 0.25))
            self.g_Cl_leak_g_val = float(g_dict.get("g_Cl_leak_g", 2.0))
        else:
            self.g_syn_bar_val, self.a_syn_val, self.T_stim_val, self.scale_stimulus = 40.0, 5e-4, 1.0, False
            self.g_Na_bar_val, self.g_K_bar_val = 1200.0, 360.0
            self.g_Na_leak_val, self.g_Na_leak_g_val = 1.0, 1.0
            self.g_K_leak_val, self.g_K_leak_g_val = 4.0, 16.96
            self.g_Cl_leak_val, self.g_Cl_leak_g_val = 0.25, 0.50

        if "stimulus_region" in config:
            self.stimulus_region = True
            self.stimulus_region_range = np.array(config["stimulus_region"]["range"]) * self.mesh_conversion_factor
            axes = {"x": 0, "y": 1, "z": 2}
            if config["stimulus_region"].get("multiple", False):
                self.multiple_stimulus_directions = True
                self.stimulus_region_directions = [axes[str(d)] for d in config["stimulus_region"]["direction"]]
            else:
                self.multiple_stimulus_directions = False
                self.stimulus_region_direction = axes[str(config["stimulus_region"]["direction"])]
        else:
            self.stimulus_region = False
            self.multiple_stimulus_directions = False

        if "initial_conditions" in config:
            self.initial_conditions = config["initial_conditions"]
            self.find_initial_conditions = False
        elif self.MMS_test:
            self.initial_conditions = {}
            self.find_initial_conditions = False
        else:
            raise NotImplementedError("Configs without 'initial_conditions' need the reference's 0-D ODE pre-processor "
                                      "(membrane_ODE_systems.py), which is out of scope of the native hot path.")
        if "membrane_data_tag" in config:
            self.membrane_data_tag = int(config["membrane_data_tag"])
        else:
            self.membrane_data_tag = self.stimulus_tags[0] if len(self.stimulus_tags) > 0 else self.gamma_tags[0]

    def parse_tags(self, tags: dict):
        allowed = {"intra", "extra", "membrane", "boundary", "glia", "neuron"}
        if not set(tags.keys()).issubset(allowed):
            raise ValueError(f"Mismatch in tags.\nAllowed tags: {allowed}\nInput tags: {set(tags.keys())}")
        if isinstance(tags["intra"], collections.abc.Sequence):
            self.print(f"# Cell tags = {len(tags['intra'])}.")
        else:
            self.print("Single cell tag.")
        self.intra_tags = tags["intra"]
        self.extra_tag = tags.get("extra", 1)
        self.gamma_tags = tags.get("membrane", self.intra_tags)
        if "glia" in tags:
            self.glia_tags = tags["glia"]
            self.glia_flag = len(self.glia_tags) > 0
        else:
            self.glia_tags, self.glia_flag = None, False
        self.neuron_tags = tags["neuron"]
        self.boundary_tags = tags.get("boundary", 1)
        self.intra_tags = tuple(_aslist(self.intra_tags))
        self.extra_tag = tuple(_aslist(self.extra_tag))
        self.boundary_tags = tuple(_aslist(self.boundary_tags))
        self.gamma_tags = tuple(_aslist(self.gamma_tags))
        self.neuron_tags = tuple(_aslist(self.neuron_tags))
        self.stimulus_tags = tuple(_aslist(self.stimulus_tags))
        if self.glia_flag:
            self.glia_tags = tuple(_aslist(self.glia_tags))

    def init_ionic_models(self, ionic_models):
        """reference mixed_dim_problem.py:435-465"""
        from .ionic_models import HodgkinHuxley, IonicModel
        if isinstance(ionic_models, IonicModel):
            ionic_models = [ionic_models]
        self.ionic_models = ionic_models
        self.gating_variables = False
        ionic_tags = set()
        for model in self.ionic_models:
            model._init()
            for tag in model.tags:
                ionic_tags.add(tag)
            self.print("Added tags for ionic model: ", str(model))
            if isinstance(model, HodgkinHuxley):
                self.gating_variables = True
                self.print("Gating variables flag set to True.")
        ionic_tags = sorted(ionic_tags)
        gamma_tags = sorted(flatten_list([self.gamma_tags]))
        if ionic_tags != gamma_tags and not self.MMS_test and len(ionic_tags) != 0:
            raise RuntimeError("Mismatch between membrane tags and ionic models tags."
                               + f"\nIonic models tags: {ionic_tags}\nMembrane tags: {gamma_tags}")
        self.print("# Membrane tags = ", len(gamma_tags))
        self.print("# Ionic models  = ", len(self.ionic_models), "\n")

    # ---------------------------------------------------------------- domain (:634-733)
    def setup_domain(self):
        self.print("Setting up mesh ...")
        if self._local_mesh_override is not None:
            lm = self._local_mesh_override
            self.mesh_description = lm.description
        elif self.MMS_test:
            # mixed_dim_problem.py:683-700: unit square / cube, one membrane tag per side of the inner box, boundary tag 8
            if self.comm.size != 1:
                raise NotImplementedError("MMS runs are single-GPU verification runs.")
            gen = meshmod.create_unit_square if self.dim == 2 else meshmod.create_unit_cube
            coords, cells = gen(self.N_mesh)
            cell_tags = meshmod.mark_subdomains_box(coords, cells)
            self.gamma_tags = (1, 2, 3, 4) if self.dim == 2 else (1, 2, 3, 4, 5, 6)
            gamma, _, gverts = meshmod.gamma_integration_entities(cells, cell_tags, self.intra_tags, self.extra_tag, None)
            cen = coords[gverts].mean(axis=1)
            gtags = np.zeros(len(gamma), dtype=np.int32)
            # misc.py:196-254 (square: LEFT 1, RIGHT 2, BOTTOM 3, TOP 4) / :400-503 (cube: LEFT, RIGHT, FRONT, BACK, BOTTOM, TOP)
            sides = [(0, 0.25, 1), (0, 0.75, 2), (1, 0.25, 3), (1, 0.75, 4)] if self.dim == 2 else \
                    [(0, 0.25, 1), (0, 0.75, 2), (1, 0.25, 3), (1, 0.75, 4), (2, 0.25, 5), (2, 0.75, 6)]
            for ax, val, tag in sides:
                gtags[np.isclose(cen[:, ax], val)] = tag
            assert gtags.min() >= 1, "MMS meshes need N_mesh divisible by 4"
            lm = partition_mesh(coords, cells, cell_tags, gamma, gtags, 1, 0)
            lm.description = f"MMS unit {'square' if self.dim == 2 else 'cube'} N={self.N_mesh}"
            self.mesh_description = lm.description
        else:
            coords, cells, cell_tags, facet_tags, desc = meshmod.load_mesh(
                self.input_files["mesh_file"], self.input_files["facet_file"], self.mesh_conversion_factor)
            self.mesh_description = desc
            if np.all(np.array(self.intra_tags) < np.array(self.extra_tag).min()) or \
               np.all(np.array(self.intra_tags) > np.array(self.extra_tag).max()):
                pass
            else:
                raise RuntimeError("Intracellular tags must be all smaller or all larger than extracellular tag.")
            gamma, gtags, gverts = meshmod.gamma_integration_entities(cells, cell_tags, self.intra_tags, self.extra_tag, facet_tags)
            keep = np.isin(gtags, self.gamma_tags)
            gamma, gtags = gamma[keep], gtags[keep]
            # meshes read from files (reconstructions) are cut by the weighted graph partitioner, the generated lattices geometrically
            generated = any(str(self.input_files["mesh_file"]).split("/")[-1].startswith(pfx) for pfx in ("square", "cube", "tissue"))
            lm = partition_mesh(coords, cells, cell_tags, gamma, gtags, self.comm.size, self.comm.rank, intra_tags=self.intra_tags,
                                method=self.config.get("partition", "rcb" if generated else "kway") if hasattr(self, "config") else None)
            lm.description = desc
        # optional renumbering along a space-filling curve (config key ``vertex_order`` / KNP_VERTEX_ORDER: native | morton)
        order = os.environ.get("KNP_VERTEX_ORDER") or (self.config.get("vertex_order", "native") if hasattr(self, "config") else "native")
        if order not in ("native", "none", ""):
            from .parallel import reorder_local_mesh
            lm = reorder_local_mesh(lm, order)
            self.mesh_description = lm.description
        self.local_mesh = lm
        self.mesh = meshmod.Mesh(lm.coords, lm.cells)
        self.mesh.comm = self.comm
        self.subdomains = meshmod.MeshTags(self.mesh.topology.dim, np.arange(lm.cells.shape[0]), lm.cell_tags, "ct")
        self.gamma_entities = lm.gamma
        self.gamma_facet_tags = lm.gamma_tags
        self.boundaries = meshmod.MeshTags(self.mesh.topology.dim - 1, np.arange(lm.gamma.shape[0]), lm.gamma_tags, "ft")
        self.cell_side = np.where(np.isin(lm.cell_tags, self.intra_tags), 0, 1).astype(np.uint8)
        known = np.isin(lm.cell_tags, self.intra_tags) | np.isin(lm.cell_tags, self.extra_tag)
        if not known.all():
            raise RuntimeError("Mesh contains cell tags that are neither ics_tags nor ecs_tags.")
        self.dx = _Measure("dx")
        self.dS = _Measure("dS")
        self.q_pts, self.q_w = meshmod.facet_quadrature(self.mesh.geometry.dim, 10)
        # facet vertices / measures on the host for setup-time membrane integrals
        d = self.mesh.geometry.dim
        loc = np.array([[a for a in range(d + 1) if a != lf] for lf in range(d + 1)])
        if lm.gamma.shape[0]:
            self._fv = lm.cells[lm.gamma[:, 0][:, None], loc[lm.gamma[:, 1]]]
            X = lm.coords[self._fv]
            if d == 2:
                self._fmeas = np.linalg.norm(X[:, 1] - X[:, 0], axis=1)
            else:
                self._fmeas = 0.5 * np.linalg.norm(np.cross(X[:, 1] - X[:, 0], X[:, 2] - X[:, 0]), axis=1)
        else:
            self._fv = np.zeros((0, d), dtype=np.int32)
            self._fmeas = np.zeros(0)
        self.neuron_cells = np.nonzero(np.isin(lm.cell_tags, self.neuron_tags))[0]
        if self.glia_flag:
            self.glia_cells = np.nonzero(np.isin(lm.cell_tags, self.glia_tags))[0]

    def integrate_over_membrane(self, integrand, tags):
        """assemble_scalar(integrand * dS(tags)) summed over ranks (setup-time only; each membrane
        facet is counted by the rank owning its first vertex)."""
        sel = np.isin(self.gamma_facet_tags, tags) & (self._fv[:, 0] < self.local_mesh.n_vertices_owned if len(self._fv) else np.zeros(0, bool))
        fv = self._fv[sel]
        if isinstance(integrand, (int, float)):
            val = float(integrand) * float(self._fmeas[sel].sum())
        else:
            # in blocks of facets (10^7 facets x 36 quadrature points x 3 coordinates would be 10 GB at once); the quadrature points as one
            # batched matrix product
            val, fm, x = 0.0, self._fmeas[sel], self.mesh.geometry.x
            for lo in range(0, fv.shape[0], 1 << 19):
                X = x[fv[lo:lo + (1 << 19)]]                                  # (n, d, dim)
                xq = np.matmul(self.q_pts[None, :, :], X)                     # (n, q, dim)
                env = {"x": [xq[:, :, k] for k in range(xq.shape[2])], "fields": {}}
                vals = np.broadcast_to(fem.evaluate_numpy(integrand, env), xq.shape[:2])
                val += float((fm[lo:lo + (1 << 19)][:, None] * vals * self.q_w[None, :]).sum())
        return self.comm.allreduce_sum(val)

    # ---------------------------------------------------------------- backend plumbing
    def create_backend(self):
        """Create the libknpemi_hip context for this rank's mesh (fails loudly without a HIP device)."""
        if self.backend is not None:
            return self.backend
        from .backend import Backend
        self.backend = Backend(self)
        return self.backend

    def backend_hh_update(self, model):
        be = self.create_backend()
        be.hh_update(self.phi_m_prev, self.n, self.m, self.h, float(self.dt.value), float(self.phi_rest.value),
                     bool(model.use_Rush_Larsen), int(model.time_steps_ODE))

    # class defaults (reference KNPEMIx_problem.py:983-997)
    mesh_conversion_factor = 1.0
    fem_order = 1
    MMS_test = False
    dirichlet_bcs = False
    pin_ecs_potential = False


def _aslist(v):
    if isinstance(v, (list, tuple, range)):
        return list(v)
    return [v]


class ProblemKNPEMI(MixedDimensionalProblem):

    def init(self):
        if self.MMS_test:
            self.setup_MMS_params()

    # ---- MMS parameters (KNPEMIx_problem.py:746-805)
    def setup_MMS_params(self):
        from .mms import ExactSolutionsKNPEMI
        self.print("Setting up MMS parameters ...")
        assert np.allclose([self.C_M.value, self.R.value, self.F.value, self.psi.value], [1.0] * 4)
        self.M = ExactSolutionsKNPEMI(self.mesh, self.t)
        self.exact_sols, self.src_terms = self.M.get_mms_terms()
        self.phi_i_init = self.exact_sols["phi_i_init"]
        self.phi_e_init = self.exact_sols["phi_e_init"]
        self.phi_m_init = self.phi_i_init - self.phi_e_init
        m = self.mesh
        self.ion_list = []
        for name, z in (("Na", 1.0), ("K", 1.0), ("Cl", -1.0)):
            self.ion_list.append({"name": name, "Di": Constant(m, 1.0), "De": Constant(m, 1.0), "z": Constant(m, z),
                                  "ki_init": self.exact_sols[f"{name}_i"], "ke_init": self.exact_sols[f"{name}_e"],
                                  "f_k_i": self.src_terms[f"f_{name}_i"], "f_k_e": self.src_terms[f"f_{name}_e"],
                                  "J_k_e": self.src_terms[f"J_{name}_e"], "f_I_m": self.src_terms[f"f_phi_{name}"],
                                  "f_i": Constant(m, 0.0), "f_e": Constant(m, 0.0),
                                  "g_leak": Constant(m, 0.0), "g_leak_g": Constant(m, 0.0)})
        self.Na, self.K, self.Cl = self.ion_list
        self.N_ions = len(self.ion_list)

    def _exact_nodal(self, expr):
        return self.M.evaluate(expr, self.mesh.geometry.x, float(self.t.value))

    # ---- spaces & restrictions (KNPEMIx_problem.py:28-94)
    def setup_spaces(self):
        self.print("Setting up function spaces ...")
        self.num_variables = self.N_ions + 1
        self.num_variables_total = 2 * self.num_variables
        self.V = FunctionSpace(self.mesh)
        self.V_list = [self.V.clone() for _ in range(self.num_variables_total)]
        self.V_list_ie = [self.V_list[:self.num_variables], self.V_list[self.num_variables:]]
        self.wh = [[Function(V) for V in self.V_list_ie[0]], [Function(V) for V in self.V_list_ie[1]]]
        self.u_out_i, self.u_out_e = [], []
        for idx, ion in enumerate(self.ion_list):
            self.wh[0][idx].name = f"{ion['name']}_i"
            self.wh[1][idx].name = f"{ion['name']}_e"
            self.u_out_i.append(self.wh[0][idx])
            self.u_out_e.append(self.wh[1][idx])
        self.wh[0][self.N_ions].name = "phi_i"
        self.wh[1][self.N_ions].name = "phi_e"
        self.u_out_i.append(self.wh[0][self.N_ions])
        self.u_out_e.append(self.wh[1][self.N_ions])
        # restricted dofs = vertices of intra / extra cells
        lm = self.local_mesh
        in_i = np.zeros(lm.coords.shape[0], dtype=bool)
        in_e = np.zeros(lm.coords.shape[0], dtype=bool)
        in_i[lm.cells[self.cell_side == 0].ravel()] = True
        in_e[lm.cells[self.cell_side == 1].ravel()] = True
        self.dofs_intra = np.nonzero(in_i)[0].astype(np.int32)
        self.dofs_extra = np.nonzero(in_e)[0].astype(np.int32)

    def setup_boundary_conditions(self):
        self.print("Setting up boundary conditions ...")
        self.bcs = []          # pure Neumann (reference default, KNPEMIx_problem.py:104,198)
        if self.MMS_test:
            # KNPEMIx_problem.py:109-134: extracellular concentrations and phi_e on the whole exterior boundary,
            # values interpolated from the exact solution when the BCs are created (t = 0)
            x = self.mesh.geometry.x
            on_bdry = np.any(np.isclose(x, 0.0) | np.isclose(x, 1.0), axis=1)
            self.bc_vertices = np.nonzero(on_bdry)[0].astype(np.int32)
            xb = x[self.bc_vertices]
            self.bc_values = [self.M.evaluate(self.exact_sols[f"{ion['name']}_e"], xb, float(self.t.value)) for ion in self.ion_list]
            self.bc_values.append(self.M.evaluate(self.exact_sols["phi_e"], xb, float(self.t.value)))
            self.bcs = [("extra", f, self.bc_vertices, self.bc_values[f]) for f in range(self.num_variables)]
        elif self.dirichlet_bcs:
            # KNPEMIx_problem.py:135-160: on the exterior boundary every field keeps its initial value (intracellular
            # fields only where an intracellular cell touches the boundary): k_init, phi_i = phi_m_init, phi_e = 0
            lm = self.local_mesh
            if str(getattr(lm, "description", "")).startswith("generated") or self.comm.size > 1:
                if not str(getattr(lm, "description", "")).startswith("generated"):
                    raise NotImplementedError("dirichlet_bcs on a partitioned mesh file: only the generated box meshes are supported")
                mm = self.get_min_and_max_coordinates()
                x = lm.coords
                scale = max(mm[2 * a + 1] - mm[2 * a] for a in range(x.shape[1]))
                on = np.zeros(x.shape[0], dtype=bool)
                for a in range(x.shape[1]):
                    on |= (np.abs(x[:, a] - mm[2 * a]) <= 1e-12 * scale) | (np.abs(x[:, a] - mm[2 * a + 1]) <= 1e-12 * scale)
                self.bc_vertices = np.nonzero(on)[0].astype(np.int32)
            else:
                fverts, _, _ = meshmod.exterior_facets(lm.cells)
                self.bc_vertices = np.unique(fverts).astype(np.int32)
            nb = len(self.bc_vertices)
            for side, sfx, phi0 in (("intra", "i", float(self.phi_m_init.value)), ("extra", "e", 0.0)):
                for f, ion in enumerate(self.ion_list):
                    self.bcs.append((side, f, self.bc_vertices, np.full(nb, float(ion[f"k{sfx}_init"].value))))
                self.bcs.append((side, self.N_ions, self.bc_vertices, np.full(nb, phi0)))

        elif self.pin_ecs_potential:
            # KNPEMIx_problem.py:163-196: phi_e = 0 at one vertex that is not on a membrane.  The reference starts from
            # vertex 0 and falls back to random picks; here: the non-membrane extracellular vertex of lowest global id.
            lm = self.local_mesh
            on_gamma = np.zeros(lm.coords.shape[0], dtype=bool)
            if lm.gamma.shape[0]:
                on_gamma[np.unique(self._fv)] = True
            is_e = np.zeros(lm.coords.shape[0], dtype=bool)
            is_e[lm.cells[self.cell_side == 1].ravel()] = True
            cand = np.nonzero(is_e & ~on_gamma & (np.arange(lm.coords.shape[0]) < lm.n_vertices_owned))[0]
            mine = int(lm.l2g[cand].min()) if cand.size else np.iinfo(np.int64).max
            best = int(-self.comm.allreduce_max(-float(mine)))
            verts = cand[lm.l2g[cand] == best].astype(np.int32) if mine == best else np.zeros(0, dtype=np.int32)
            self.bc_vertices = verts
            self.bcs = [("extra", self.N_ions, verts, np.zeros(len(verts)))]
            self.print(f"Phi_e pinned at global vertex {best}")

    # ---- ion injection (mixed_dim_problem.py:467-541, 806-811; KNPEMIx_problem.py:200-218)
    def get_min_and_max_coordinates(self):
        x = self.local_mesh.coords
        out = []
        for a in range(x.shape[1]):
            out += [-self.comm.allreduce_max(-x[:, a].min()), self.comm.allreduce_max(x[:, a].max())]
        return out

    def calculate_mesh_center(self):
        mm = self.get_min_and_max_coordinates()
        c = [(mm[2 * a] + mm[2 * a + 1]) / 2 for a in range(len(mm) // 2)]
        return np.array(c + [0.0] * (3 - len(c)))

    def initialize_injection_site(self, delta: float):
        """Cells with every vertex inside the cube of half-width delta around the mesh centre, and their volume."""
        lm = self.local_mesh
        x = np.zeros((lm.coords.shape[0], 3))
        x[:, :lm.coords.shape[1]] = lm.coords
        c = self.calculate_mesh_center()
        self.x_L, self.y_L, self.z_L = c - delta
        self.x_U, self.y_U, self.z_U = c + delta
        tol = 1e-14
        inside = ((x >= (c - delta) - tol) & (x <= (c + delta) + tol)).all(axis=1)
        self.injection_cells = np.nonzero(inside[lm.cells].all(axis=1))[0]
        d = lm.coords.shape[1]
        X = lm.coords[lm.cells]
        vol = np.abs(np.linalg.det(X[:, 1:, :] - X[:, :1, :])) / math.factorial(d)
        owned = self.injection_cells[self.injection_cells < lm.n_cells_owned]
        self.injection_volume = self.comm.allreduce_sum(float(vol[owned].sum()))

    def setup_source_terms(self):
        """K and Cl are injected into the extracellular space at 5 nA (KNPEMIx_problem.py:200-218)."""
        mm = self.get_min_and_max_coordinates()
        self.initialize_injection_site(delta=(mm[1] - mm[0]) / 10)
        if not self.injection_volume > 0:
            raise RuntimeError("ion_injection: no mesh cell lies inside the injection site")
        I = 5e-9
        src_term = I / (1 * float(self.F.value)) / self.injection_volume
        verts = np.unique(self.local_mesh.cells[self.injection_cells])
        for idx in (1, 2):
            f = fem.Function(self.V, name=f"f_e_{self.ion_list[idx]['name']}")
            arr = np.zeros(self.local_mesh.coords.shape[0])
            arr[verts] = src_term
            f.x.array[:] = torch.as_tensor(arr, dtype=f.x.array.dtype, device=f.x.array.device)
            self.ion_list[idx]["f_e"] = f
        if getattr(self, "backend", None) is not None:
            self.backend.set_sources()

    # ---- initial conditions (KNPEMIx_problem.py:326-353, 386-452)
    def set_initial_conditions(self):
        if self.MMS_test:
            # KNPEMIx_problem.py:363-385, 417-431: interpolate the exact solution at t = 0
            self.print("Setting initial conditions for MMS test ...")
            dev = self.mesh.device
            as_t = lambda a: torch.as_tensor(np.array(a, dtype=np.float64, copy=True), device=dev)
            self.phi_m_prev = Function(self.V, "phi_m")
            self.phi_m_prev.x.array[:] = as_t(self._exact_nodal(self.phi_m_init))
            self.wh[0][self.N_ions].x.array[:] = as_t(self._exact_nodal(self.phi_i_init))
            self.wh[1][self.N_ions].x.array[:] = as_t(self._exact_nodal(self.phi_e_init))
            for idx, ion in enumerate(self.ion_list):
                self.wh[0][idx].x.array[:] = as_t(self._exact_nodal(ion["ki_init"]))
                self.wh[1][idx].x.array[:] = as_t(self._exact_nodal(ion["ke_init"]))
            self.print("Initial conditions set.")
            return
        self.print("Setting initial conditions from input file ...")
        ic = self.initial_conditions
        if not self.glia_flag:
            self.phi_m_init.value = ic["phi_m"] if "phi_m" in ic else ic["phi_m_n"]
            self.Na_i_init.value = ic["Na_i"] if "Na_i" in ic else ic["Na_i_n"]
            self.Na_e_init.value = ic["Na_e"]
            self.K_i_init.value = ic["K_i"] if "K_i" in ic else ic["K_i_n"]
            self.K_e_init.value = ic["K_e"]
            self.Cl_i_init.value = ic["Cl_i"] if "Cl_i" in ic else ic["Cl_i_n"]
            self.Cl_e_init.value = ic["Cl_e"]
        else:
            self.phi_m_n_init.value = ic["phi_m_n"]; self.phi_m_g_init.value = ic["phi_m_g"]
            self.Na_i_n_init.value = ic["Na_i_n"]; self.Na_i_g_init.value = ic["Na_i_g"]
            self.Na_e_init.value = ic["Na_e"]
            self.K_i_n_init.value = ic["K_i_n"]; self.K_i_g_init.value = ic["K_i_g"]
            self.K_e_init.value = ic["K_e"]
            self.Cl_i_n_init.value = ic["Cl_i_n"]; self.Cl_i_g_init.value = ic["Cl_i_g"]
            self.Cl_e_init.value = ic["Cl_e"]
        self.n_init.value = ic["n"]; self.m_init.value = ic["m"]; self.h_init.value = ic["h"]

        self.phi_m_prev = Function(self.V, "phi_m")
        ui_p, ue_p = self.wh[0], self.wh[1]
        if not self.glia_flag:
            self.phi_m_prev.x.array[:] = self.phi_m_init.value
            self.print(f"Initial membrane potential: {self.phi_m_init.value}")
            ui_p[self.N_ions].x.array[:] = self.phi_m_init.value
            ue_p[self.N_ions].x.array[:] = 0.0
        else:
            dev = self.mesh.device
            self.neuron_dofs = torch.as_tensor(np.unique(self.local_mesh.cells[self.neuron_cells].ravel()), device=dev, dtype=torch.long)
            self.glia_dofs = torch.as_tensor(np.unique(self.local_mesh.cells[self.glia_cells].ravel()), device=dev, dtype=torch.long)
            self.phi_m_prev.x.array[self.neuron_dofs] = self.phi_m_n_init.value
            self.phi_m_prev.x.array[self.glia_dofs] = self.phi_m_g_init.value
            ui_p[self.N_ions].x.array[self.neuron_dofs] = self.phi_m_n_init.value
            ui_p[self.N_ions].x.array[self.glia_dofs] = self.phi_m_g_init.value
            ue_p[self.N_ions].x.array[:] = 0.0
        for idx, ion in enumerate(self.ion_list):
            if self.glia_flag:
                ui_p[idx].x.array[self.neuron_dofs] = ion["ki_init_n"].value
                ui_p[idx].x.array[self.glia_dofs] = ion["ki_init_g"].value
                ue_p[idx].x.array[:] = ion["ke_init"].value
            else:
                ui_p[idx].x.array[:] = ion["ki_init"].value
                ue_p[idx].x.array[:] = ion["ke_init"].value
                self.print(f"Initial condition for {ion['name']}_i set to {ion['ki_init'].value}")
                self.print(f"Initial condition for {ion['name']}_e set to {ion['ke_init'].value}")
        self.print("Initial conditions set.")

    # ---- "variational form": compile the membrane currents (KNPEMIx_problem.py:504-555)
    def setup_variational_form(self):
        from .ionic_models import HodgkinHuxley
        self.print("Setting up variational form ...")
        psi = self.psi
        ui_p, ue_p = self.wh[0], self.wh[1]
        for idx, ion in enumerate(self.ion_list):
            ion["E"] = (psi / ion["z"]) * fem.ln(ue_p[idx] / ui_p[idx])                # :516
            ion["I_ch"] = dict.fromkeys(self.gamma_tags, fem.ZeroBaseForm(None))
        I_ch = dict.fromkeys(self.gamma_tags, fem.ZeroBaseForm(None))
        # The loop is the reference's (KNPEMIx_problem.py:504-555), tag by tag.  A tissue configuration carries one membrane tag per cell
        # (10^5 of them at the size of configs[3]) with the same mechanisms on every one: a mechanism's expression for an ion -- and its
        # stimulus term -- is built once and shared by its tags (expressions are immutable trees), and ``terms`` records which shared
        # terms a tag's current consists of, so that tags with the same terms are compiled once further down.
        stim_tags = set(self.stimulus_tags)
        terms = {tag: [[] for _ in range(self.N_ions)] for tag in self.gamma_tags}
        for idx, ion in enumerate(self.ion_list):
            for model in self.ionic_models:
                base = with_stim = stim = None
                hh_na = ion["name"] == "Na" and isinstance(model, HodgkinHuxley)
                for gamma_tag in model.tags:
                    if base is None:
                        base = model._eval(idx)
                    I_ch_k_ = base
                    if hh_na and gamma_tag in stim_tags:
                        if with_stim is None:
                            if self.stimulus_region:
                                if not self.multiple_stimulus_directions:
                                    stim = model._add_stimulus(idx, step=True, range=self.stimulus_region_range, dir=self.stimulus_region_direction)
                                else:
                                    stim = model._add_stimulus(idx, step=True, range=self.stimulus_region_range, dir=self.stimulus_region_directions)
                            else:
                                stim = model._add_stimulus(idx, step=True)
                            with_stim = base + stim
                        I_ch_k_ = with_stim
                        self.print(f"Stimulus added on membrane with tag {gamma_tag}.")
                        self.stim_ufl_expr = stim
                    ion["I_ch"][gamma_tag] = ion["I_ch"][gamma_tag] + I_ch_k_
                    I_ch[gamma_tag] = I_ch[gamma_tag] + I_ch_k_
                    terms.setdefault(gamma_tag, [[] for _ in range(self.N_ions)])[idx].append(id(I_ch_k_))
        # one bytecode program per membrane tag
        roles = {}
        for j in range(self.N_ions):
            roles[id(ui_p[j])] = ("KI", j)
            roles[id(ue_p[j])] = ("KE", j)
        roles[id(self.phi_m_prev)] = ("PHIM", 0)
        self.aux_functions = []
        # Tissue configs carry hundreds of membrane tags with the same mechanism list: tags whose compiled program is
        # identical (same bytecode, same constant objects) share one program id.
        # Constants somebody holds a handle to (attributes of the problem, of a mechanism, entries of the ion table) may
        # be changed later and are compared by identity; the anonymous literals the mechanisms create inside _eval
        # (Constant(mesh, 0.0) ...) are only reachable through the expression and are compared by value.
        named = set()
        for holder in [vars(self)] + [vars(m) for m in self.ionic_models] + list(self.ion_list):
            named.update(id(v) for v in holder.values() if isinstance(v, fem.Constant))
        ckey = lambda c: (("id", id(c)) if id(c) in named else ("v", float(c.value))) if isinstance(c, fem.Constant) else ("v", float(c))
        self.programs = {}
        self.tag_program = {}
        seen = {}
        by_terms = {}      # tags whose currents are sums of the same shared terms: compiled once
        for k, tag in enumerate(self.gamma_tags):
            tkey = tuple(tuple(t) for t in terms[tag])
            if tkey in by_terms:
                self.tag_program[k] = by_terms[tkey]
                continue
            outs = [self.ion_list[j]["I_ch"][tag] for j in range(self.N_ions)]
            spec = fem.compile_program(outs, roles, self.aux_functions)
            key = (spec.code.tobytes(), tuple(ckey(c) for c in spec.const_sources))
            if key not in seen:
                seen[key] = len(self.programs)
                self.programs[seen[key]] = spec
            self.tag_program[k] = by_terms[tkey] = seen[key]
        if self.backend is not None and getattr(self.backend, "tag_program", None) != self.tag_program:
            raise RuntimeError("setup_variational_form() changed the membrane-tag -> program map after the backend was created")
        self.a = "hard-wired in knp_kernels.hip (k_assemble_pairs, k_gamma_facets, k_gamma_pairs)"
        self.L = "hard-wired in knp_kernels.hip (k_rhs, k_gamma_facets) + membrane programs"
        if self.backend is not None:
            self.backend.upload_programs()

    def setup_preconditioner(self, use_block_jacobi: bool):
        self.print("Setting up preconditioner ...")
        # use_block_jacobi=False (KNPEMIx_problem.py:720-722): the phi rows of P keep the -D grad k flux, i.e. P additionally has
        # the (phi,k) blocks dt z_j D_j K -- exactly the (phi,k) blocks of A.  The diagonal blocks are the same either way; the
        # solver then applies P as a block forward substitution (knp_pc_setup kind KNP_PC_AMG_LT) instead of block-diagonally.
        self.P_block_jacobi = bool(use_block_jacobi)
        self.P = "hard-wired in knp_kernels.hip (k_assemble_nodes<true>, k_gamma_pairs<true>)" + ("" if use_block_jacobi else " + (phi,k) blocks of A")

    def print_conservation(self):
        be = self.create_backend()
        tot = be.total_ion_amounts()
        self.print(f"Time {self.t.value*1e3:.2f} ms")
        for name, v in zip(("Na+", "K+ ", "Cl-"), tot):
            self.print(f"Total {name} concentration: {v:.2e} mol")

    def print_errors(self):
        """L2 errors against the exact solution at the current time (KNPEMIx_problem.py:845-907)."""
        if not hasattr(self, "_mms_asm"):
            from .mms import MMSAssembler
            self._mms_asm = MMSAssembler(self)
        e = self._mms_asm.l2_errors()          # [Na_i, Na_e, K_i, K_e, Cl_i, Cl_e, phi_i, phi_e]
        self.print("#-------------- ERRORS --------------#")
        for nm, v in zip(("Na_i ", "Na_e ", "K_i  ", "K_e  ", "Cl_i ", "Cl_e ", "phi_i", "phi_e"), e):
            self.print(f"L2 {nm} error:", v)
        self.errors = e

    # ---- constants (KNPEMIx_problem.py:909-981)
    def setup_constants(self):
        m = self.mesh
        self.C_M = Constant(m, self.C_M_value)
        self.T = Constant(m, self.T_value)
        self.F = Constant(m, self.F_value)
        self.R = Constant(m, self.R_value)
        self.psi = Constant(m, self.psi_value)
        self.g_Na_bar = Constant(m, self.g_Na_bar_val)
        self.g_K_bar = Constant(m, self.g_K_bar_val)
        self.g_Na_leak = Constant(m, self.g_Na_leak_val)
        self.g_Na_leak_g = Constant(m, self.g_Na_leak_g_val)
        self.g_K_leak = Constant(m, self.g_K_leak_val)
        self.g_K_leak_g = Constant(m, self.g_K_leak_g_val)
        self.g_Cl_leak = Constant(m, self.g_Cl_leak_val)
        self.g_Cl_leak_g = Constant(m, self.g_Cl_leak_g_val)
        self.g_syn_bar = Constant(m, self.g_syn_bar_val)
        self.a_syn = Constant(m, self.a_syn_val)
        self.T_stim = Constant(m, self.T_stim_val)
        self.D_Na = Constant(m, 1.33e-9)
        self.D_K = Constant(m, 1.96e-9)
        self.D_Cl = Constant(m, 2.03e-9)
        self.phi_rest = Constant(m, -0.065)
        self.rho_pump = Constant(m, 1.115e-6)
        self.P_Nai = Constant(m, 10)
        self.P_Ke = Constant(m, 1.5)
        self.k_dec = Constant(m, 2.9e-8)
        self.phi_m_init = Constant(m, -0.070)
        self.Na_i_init = Constant(m, 10); self.Na_e_init = Constant(m, 145)
        self.K_i_init = Constant(m, 130); self.K_e_init = Constant(m, 3)
        self.Cl_i_init = Constant(m, 5); self.Cl_e_init = Constant(m, 134)
        self.phi_m_n_init = Constant(m, self.phi_m_init.value)
        self.phi_m_g_init = Constant(m, -0.085)
        self.Na_i_n_init = Constant(m, self.Na_i_init.value)
        self.K_i_n_init = Constant(m, self.K_i_init.value)
        self.Cl_i_n_init = Constant(m, self.Cl_i_init.value)
        self.Na_i_g_init = Constant(m, 15)
        self.K_i_g_init = Constant(m, 100)
        self.Cl_i_g_init = Constant(m, 5)
        self.n_init = Constant(m, 0.24458654944007155)
        self.m_init = Constant(m, 0.028905534475191896)
        self.h_init = Constant(m, 0.7540796658225248)
        self.Na_e_f = Constant(m, 0.0); self.Na_i_f = Constant(m, 0.0)
        self.K_e_f = Constant(m, 0.0); self.K_i_f = Constant(m, 0.0)
        self.Cl_e_f = Constant(m, 0.0); self.Cl_i_f = Constant(m, 0.0)
        self.Na = {"name": "Na", "g_leak": self.g_Na_leak, "g_leak_g": self.g_Na_leak_g, "Di": self.D_Na, "De": self.D_Na,
                   "ki_init": self.Na_i_init, "ke_init": self.Na_e_init, "ki_init_n": self.Na_i_n_init,
                   "ki_init_g": self.Na_i_g_init, "z": Constant(m, 1.0), "f_e": self.Na_e_f, "f_i": self.Na_i_f}
        self.K = {"name": "K", "g_leak": self.g_K_leak, "g_leak_g": self.g_K_leak_g, "Di": self.D_K, "De": self.D_K,
                  "ki_init": self.K_i_init, "ke_init": self.K_e_init, "ki_init_n": self.K_i_n_init,
                  "ki_init_g": self.K_i_g_init, "z": Constant(m, 1.0), "f_e": self.K_e_f, "f_i": self.K_i_f}
        self.Cl = {"name": "Cl", "g_leak": self.g_Cl_leak, "g_leak_g": self.g_Cl_leak_g, "Di": self.D_Cl, "De": self.D_Cl,
                   "ki_init": self.Cl_i_init, "ke_init": self.Cl_e_init, "ki_init_n": self.Cl_i_n_init,
                   "ki_init_g": self.Cl_i_g_init, "z": Constant(m, -1.0), "f_e": self.Cl_e_f, "f_i": self.Cl_i_f}
        self.ion_list = [self.Na, self.K, self.Cl]
        self.N_ions = len(self.ion_list)
