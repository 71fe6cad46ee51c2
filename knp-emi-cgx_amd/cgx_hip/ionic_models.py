"""Membrane-mechanism plugin surface of the reference, on top of the array expression layer.

Same class names, constructor signatures and ``_init`` / ``_eval(ion_idx)`` / ``__str__``
contract as reference src/CGx/KNPEMI/KNPEMIx_ionic_model.py, so user-defined mechanisms written
against the reference keep working: ``_eval`` returns an expression built from
``problem.wh[..]``, ``problem.phi_m_prev``, gating Functions and Constants; the problem compiles
it to the bytecode run by the HIP membrane kernel at every Gamma quadrature point.

The Hodgkin-Huxley gating ODE (reference :605-671) runs in the HIP kernel ``k_hh_update``.
"""
from __future__ import annotations

import math
import time
from abc import ABC, abstractmethod

import numpy as np

from . import fem as ufl
from .fem import Constant, Function


class IonicModel(ABC):
    """Base class (reference :11-75)."""

    def __init__(self, KNPEMIx_problem, tags: tuple = None):
        self.problem = KNPEMIx_problem
        self.tags = tags
        if self.tags is None:
            self.tags = self.problem.gamma_tags
        if isinstance(self.tags, (int, np.integer)):
            self.tags = (int(self.tags),)
        self.zero = Constant(self.problem.mesh, 0.0)

    @abstractmethod
    def _init(self):
        pass

    @abstractmethod
    def _eval(self, ion_idx):
        pass

    def f_NKCC1(self, K_e, K_e_0, K_min_val: float = 3.0, eps: float = 1e-6, cap: float = 1.0):
        """NKCC1 silencing factor.  The reference tests a *symbolic* conditional with a Python
        ``if`` (:62-69); a symbolic expression is always truthy, so the band test always takes the
        first branch and the factor is identically zero.  Reproduced as is (the expression layer
        keeps UFL's truthiness) so that results match the reference."""
        K_min = Constant(self.problem.mesh, K_min_val)
        if ufl.conditional(ufl.Or(ufl.lt(K_e, K_min), ufl.gt(K_e, K_e_0)), True, False):
            return self.zero
        denom = ufl.max_value(K_e - K_e_0, eps)
        val = 1.0 / (1.0 + (0.03 / denom) ** 10)
        return ufl.min_value(ufl.max_value(val, self.zero), cap)


class PassiveModel(IonicModel):
    """I_ch^k = phi_m for every ion (reference :77-91)."""

    def _init(self):
        pass

    def __str__(self):
        return "Passive model"

    def _eval(self, ion_idx: int):
        return self.problem.phi_m_prev


class KirNaKPumpModel(IonicModel):
    """Glial Kir4.1 current + Na/K-ATPase (reference :93-222)."""

    rho_pump_val = 1.1 * 1.12e-6
    P_Na_i_val = 10.0
    P_K_e_val = 1.5

    def __init__(self, KNPEMIx_problem, tags: tuple = None):
        super().__init__(KNPEMIx_problem, tags)
        p = KNPEMIx_problem
        self.E_K_init = Constant(p.mesh, p.psi.value * np.log(p.K_e_init.value / p.K_i_g_init.value))
        self.rho_pump = Constant(p.mesh, self.rho_pump_val)
        self.P_Na_i = Constant(p.mesh, self.P_Na_i_val)
        self.P_K_e = Constant(p.mesh, self.P_K_e_val)

    def __str__(self):
        return "Na/K/ATPase pump with passive inward-rectifying K current"

    def _init(self):
        p = self.problem
        c_Na_i = p.wh[0][0]
        c_K_e = p.wh[1][1]
        self.pump_coeff = ((1.0 / (1.0 + (self.P_Na_i / c_Na_i) ** (3 / 2)))
                           * (1.0 / (1.0 + self.P_K_e / c_K_e)) * self.rho_pump)

    def _eval(self, ion_idx: int):
        p = self.problem
        ion = p.ion_list[ion_idx]
        phi_m = p.phi_m_prev
        F, z = p.F, ion["z"]
        if ion["name"] == "K":
            delta_phi = phi_m - ion["E"]
            f_kir = self.f_Kir(p.K_e_init, p.wh[1][ion_idx], self.E_K_init, delta_phi, phi_m)
            I_ATP = -2 * z * F * self.pump_coeff
        else:
            f_kir = Constant(p.mesh, 1.0)
            I_ATP = 3 * z * F * self.pump_coeff if ion["name"] == "Na" else Constant(p.mesh, 0.0)
        I_kir = f_kir * ion["g_leak_g"] * (phi_m - ion["E"])
        return I_kir + I_ATP

    def f_Kir(self, K_e_init, K_e, E_K_init, delta_phi, phi_m):
        A = 1 + math.exp(0.433)
        B = 1 + ufl.exp(-(0.1186 + E_K_init) / 0.0441)
        C = 1 + ufl.exp((delta_phi + 0.0185) / 0.0425)
        D = 1 + ufl.exp(-(0.1186 + phi_m) / 0.0441)
        return ufl.sqrt(K_e / K_e_init) * A * B / (C * D)


class _Cotransporters(IonicModel):
    def _concentrations(self):
        p = self.problem
        return (p.wh[0][0], p.wh[1][0], p.wh[0][1], p.wh[1][1], p.wh[0][2], p.wh[1][2], p.K_e_init)


class GlialCotransporters(_Cotransporters):
    """KCC1 / NKCC1 (reference :224-298)."""

    def __str__(self):
        return "KCC1/NKCC1 Cotransporters"

    def _init(self):
        p = self.problem
        self.S_KCC1 = Constant(p.mesh, 7e-2 * p.psi.value)
        self.S_NKCC1 = Constant(p.mesh, 2e-2 * p.psi.value)

    def _eval(self, ion_idx: int):
        ion = self.problem.ion_list[ion_idx]
        Na_i, Na_e, K_i, K_e, Cl_i, Cl_e, K_e_0 = self._concentrations()
        I_KCC1 = self.S_KCC1 * ufl.ln((K_i * Cl_i) / (K_e * Cl_e))
        I_NKCC1 = self.S_NKCC1 * self.f_NKCC1(K_e, K_e_0) * ufl.ln((Na_e * K_e * Cl_e ** 2) / (Na_i * K_i * Cl_i ** 2))
        if ion["name"] == "Na":
            return -I_NKCC1
        if ion["name"] == "K":
            return -I_NKCC1 + I_KCC1
        return 2 * I_NKCC1 - I_KCC1


class NeuronalCotransporters(_Cotransporters):
    """KCC2 / NKCC1 (reference :300-369)."""

    def __str__(self):
        return "KCC2/NKCC1 Cotransporters"

    def _init(self):
        self.S_KCC2 = Constant(self.problem.mesh, 0.0068)
        self.S_NKCC1 = Constant(self.problem.mesh, 0.0023)

    def _eval(self, ion_idx: int):
        ion = self.problem.ion_list[ion_idx]
        Na_i, Na_e, K_i, K_e, Cl_i, Cl_e, K_e_0 = self._concentrations()
        I_KCC2 = self.S_KCC2 * ufl.ln((K_i * Cl_i) / (K_e * Cl_e))
        I_NKCC1 = self.S_NKCC1 * self.f_NKCC1(K_e, K_e_0) * ufl.ln((Na_e * K_e * Cl_e ** 2) / (Na_i * K_i * Cl_i ** 2))
        if ion["name"] == "Na":
            return -I_NKCC1
        if ion["name"] == "K":
            return -I_NKCC1 + I_KCC2
        return I_NKCC1 - I_KCC2


class ATPPump(IonicModel):
    """Neuronal Na/K-ATPase (reference :371-424)."""

    def __str__(self):
        return "Na/K/ATPase pump"

    def _init(self):
        m = self.problem.mesh
        self.I_hat = Constant(m, 0.25)
        self.P_K_e = Constant(m, 1.5)
        self.P_Na_i = Constant(m, 10.0)

    def _eval(self, ion_idx: int):
        p = self.problem
        ion = p.ion_list[ion_idx]
        if ion["name"] == "Cl":
            return Constant(p.mesh, 0.0)
        par_1 = 1 + self.P_K_e / p.wh[1][1]
        par_2 = 1 + self.P_Na_i / p.wh[0][0]
        I_ATP = self.I_hat / (par_1 ** 2 * par_2 ** 3)
        if ion["name"] == "Na":
            return 3 * I_ATP
        if ion["name"] == "K":
            return -2 * I_ATP
        raise ValueError("Unknown ion for ATP pump model.")


class HodgkinHuxley(IonicModel):
    """Hodgkin-Huxley channels with optional synaptic stimulus (reference :426-674)."""

    def __init__(self, KNPEMIx_problem, tags: tuple = None, use_Rush_Larsen: bool = True, time_steps_ODE: int = 25):
        super().__init__(KNPEMIx_problem, tags)
        self.use_Rush_Larsen = use_Rush_Larsen
        self.time_steps_ODE = time_steps_ODE
        self.dt_ode = KNPEMIx_problem.dt.value / self.time_steps_ODE
        self.T_stim = KNPEMIx_problem.T_stim.value
        if hasattr(KNPEMIx_problem, "tau_syn_rise"):
            self.tau_syn_rise = Constant(KNPEMIx_problem.mesh, KNPEMIx_problem.tau_syn_rise)
            self.tau_syn_decay = Constant(KNPEMIx_problem.mesh, KNPEMIx_problem.tau_syn_decay)

    def __str__(self):
        return "Hodgkin-Huxley"

    def _init(self):
        p = self.problem
        p.n = Function(p.V, "n")
        p.m = Function(p.V, "m")
        p.h = Function(p.V, "h")
        p.n.x.array[:] = p.n_init.value
        p.m.x.array[:] = p.m_init.value
        p.h.x.array[:] = p.h_init.value
        p.print(f"Initial n = {p.n_init.value}\nm = {p.m_init.value}\nh = {p.h_init.value}")
        self.t_mod = Constant(p.mesh, 0.0)

    def _eval(self, ion_idx: int):
        p = self.problem
        ion = p.ion_list[ion_idx]
        g_k = ion["g_leak"]
        if ion["name"] == "Na":
            g_k = g_k + p.g_Na_bar * p.m ** 3 * p.h
        elif ion["name"] == "K":
            g_k = g_k + p.g_K_bar * p.n ** 4
        return g_k * (p.phi_m_prev - ion["E"])

    def _add_stimulus(self, ion_idx: int, step: bool, range=None, dir=None):
        p = self.problem
        ion = p.ion_list[ion_idx]
        assert ion["name"] == "Na", "Only Na can have a stimulus current in the Hodgkin-Huxley model."
        if step:
            exp_factor = ufl.exp(-self.t_mod / p.a_syn)
        else:
            exp_factor = ufl.exp(-self.t_mod / self.tau_syn_decay) - ufl.exp(-self.t_mod / self.tau_syn_rise)
        if range is None:
            mask = 1.0
        else:
            x = ufl.SpatialCoordinate(p.mesh)
            if not p.multiple_stimulus_directions:
                mask = ufl.conditional(ufl.And(ufl.gt(x[dir], float(range[0])), ufl.lt(x[dir], float(range[1]))), 1.0, 0.0)
            else:
                mask = 1.0
                for i, d in enumerate(dir):
                    mask = mask * ufl.conditional(ufl.And(ufl.gt(x[d], float(range[i][0])), ufl.lt(x[d], float(range[i][1]))), 1.0, 0.0)
        stim_current = mask * p.g_syn_bar * exp_factor * (p.phi_m_prev - ion["E"])
        if p.scale_stimulus:
            # the reference integrates the mask over all stimulus tags once per membrane tag; the value only depends on
            # the region, so tissue configs with hundreds of tags integrate once
            key = (None if range is None else tuple(np.ravel(np.asarray(range, dtype=float)).tolist()),
                   None if dir is None else tuple(np.ravel(dir).tolist()), tuple(p.stimulus_tags))
            cache = p.__dict__.setdefault("_stimulus_area_cache", {})
            if key not in cache:
                cache[key] = p.integrate_over_membrane(mask, p.stimulus_tags)
            p.stimulus_area = cache[key]
            p.print(f"Stimulus area on tag {p.stimulus_tags[0]}: {p.stimulus_area:0.6e} m^2")
            stim_current = stim_current * (1.0 / p.stimulus_area)
        return stim_current

    def update_gating_variables(self):
        """n, m, h <- one PDE step of the gating ODEs, frozen rates (reference :605-671), on the GPU."""
        tic = time.perf_counter()
        p = self.problem
        p.backend_hh_update(self)
        self.last_ode_time = time.perf_counter() - tic

    def update_t_mod(self, tol: float = 1e-12):
        self.t_mod.value = np.mod(self.problem.t.value + tol, self.T_stim)
