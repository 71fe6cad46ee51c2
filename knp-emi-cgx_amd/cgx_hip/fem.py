"""Light array-backed stand-ins for the DOLFINx / UFL objects that the reference's problem and
membrane-mechanism code manipulates (dfx.fem.Function, dfx.fem.Constant, UFL expressions), plus the
compiler that turns a mechanism's ``_eval`` expression into the register bytecode interpreted by
the HIP membrane kernel (include/knpemi_hip.h, KNP_OP_*).

Replaces: UFL expression building + FFCx JIT of the Gamma integrands
(reference src/CGx/KNPEMI/KNPEMIx_problem.py:504-555,609-610,641-642, 654-655).

Semantics kept from UFL on purpose:
  * ``bool(expr)`` is True for every expression (UFL ``Expr.__bool__``) -- this is what makes the
    reference's ``IonicModel.f_NKCC1`` return zero (KNPEMIx_ionic_model.py:62-69).
  * ``expr('+')`` / ``expr('-')`` restrictions are accepted and are no-ops: all coefficients are
    continuous P1 fields, only facet-vertex values enter a Gamma integral.
"""
from __future__ import annotations

import math
import numbers

import numpy as np
import torch

from ._lib import KNP_MAX_AUX, KNP_MAX_PROG_REGS, OPS


# ------------------------------------------------------------------------------------------
# expressions
# ------------------------------------------------------------------------------------------
class Expr:
    __array_priority__ = 1000          # make numpy scalars defer to our operators

    def _bin(self, op, other, swap=False):
        other = as_expr(other)
        return Op(op, (other, self) if swap else (self, other))

    def __add__(self, o): return self._bin("ADD", o)
    def __radd__(self, o): return self._bin("ADD", o, True)
    def __sub__(self, o): return self._bin("SUB", o)
    def __rsub__(self, o): return self._bin("SUB", o, True)
    def __mul__(self, o): return self._bin("MUL", o)
    def __rmul__(self, o): return self._bin("MUL", o, True)
    def __truediv__(self, o): return self._bin("DIV", o)
    def __rtruediv__(self, o): return self._bin("DIV", o, True)
    def __neg__(self): return Op("NEG", (self,))
    def __pos__(self): return self
    def __abs__(self): return Op("ABS", (self,))

    def __pow__(self, e):
        if isinstance(e, numbers.Integral) and not isinstance(e, bool) and abs(int(e)) <= 64:
            return Op("POWI", (self,), int(e))
        return Op("POW", (self, as_expr(e)))

    def __rpow__(self, base):
        return Op("POW", (as_expr(base), self))

    def __bool__(self):                 # UFL: every Expr is truthy
        return True

    def __call__(self, restriction=None):
        return self


class Literal(Expr):
    def __init__(self, value):
        self.value = float(value)


class Constant(Expr):
    """dfx.fem.Constant stand-in: a mutable scalar (``.value``)."""

    def __init__(self, mesh, value):
        self.mesh = mesh
        self._value = float(np.asarray(value))

    @property
    def value(self):
        return self._value

    @value.setter
    def value(self, v):
        self._value = float(np.asarray(v))

    def __float__(self):
        return self._value

    def __repr__(self):
        return f"Constant({self._value})"


class _Vector:
    """``function.x``: owns the nodal array (a torch tensor on the problem's device)."""

    def __init__(self, n, device):
        self.array = torch.zeros(n, dtype=torch.float64, device=device)

    def scatter_forward(self):          # ghosts are refreshed by the library's halo hook
        return None

    @property
    def petsc_vec(self):
        raise AttributeError("no PETSc in the MI355X-native path; use .array")


class FunctionSpace:
    def __init__(self, mesh, name="P1"):
        self.mesh = mesh
        self.name = name
        self.num_dofs = mesh.num_vertices

    def clone(self):
        return FunctionSpace(self.mesh, self.name)


class Function(Expr):
    """dfx.fem.Function stand-in for a P1 nodal field (one value per mesh vertex)."""

    def __init__(self, V, name="f"):
        self.function_space = V
        self.name = name
        self.x = _Vector(V.mesh.num_vertices, V.mesh.device)

    def numpy(self):
        return self.x.array.detach().cpu().numpy()

    def data_ptr(self):
        return self.x.array.data_ptr()


class Coordinate(Expr):
    def __init__(self, axis):
        self.axis = int(axis)


class Op(Expr):
    def __init__(self, op, args, imm=0):
        self.op, self.args, self.imm = op, tuple(args), imm


def as_expr(v):
    if isinstance(v, Expr):
        return v
    if isinstance(v, (numbers.Real, np.floating, np.integer)):
        return Literal(float(v))
    if isinstance(v, np.ndarray) and v.ndim == 0:
        return Literal(float(v))
    raise TypeError(f"cannot use {type(v)} in a membrane expression")


# UFL-named helpers (what mechanism code imports as ``ufl.<name>``)
def ln(x): return Op("LN", (as_expr(x),))
def exp(x): return Op("EXP", (as_expr(x),))
def sqrt(x): return Op("SQRT", (as_expr(x),))
def max_value(a, b): return Op("MAX", (as_expr(a), as_expr(b)))
def min_value(a, b): return Op("MIN", (as_expr(a), as_expr(b)))
def lt(a, b): return Op("LT", (as_expr(a), as_expr(b)))
def gt(a, b): return Op("GT", (as_expr(a), as_expr(b)))
def le(a, b): return Op("LE", (as_expr(a), as_expr(b)))
def ge(a, b): return Op("GE", (as_expr(a), as_expr(b)))
def eq(a, b): return Op("EQ", (as_expr(a), as_expr(b)))
def And(a, b): return Op("AND", (as_expr(a), as_expr(b)))
def Or(a, b): return Op("OR", (as_expr(a), as_expr(b)))
def Not(a): return Op("NOT", (as_expr(a),))
def conditional(c, t, f): return Op("SEL", (as_expr(c), as_expr(t), as_expr(f)))
def SpatialCoordinate(mesh): return [Coordinate(d) for d in range(mesh.geometry.dim)]


max = max_value      # noqa: A001  (the reference spells it ufl.max / ufl.min)
min = min_value      # noqa: A001


class ZeroBaseForm(Expr):
    """``ufl.ZeroBaseForm(None)``: additive identity used to start sums."""

    def __init__(self, *_):
        pass

    def __add__(self, o): return as_expr(o)
    def __radd__(self, o): return as_expr(o)


# ------------------------------------------------------------------------------------------
# compilation to bytecode
# ------------------------------------------------------------------------------------------
class ProgramSpec:
    """Compiled membrane program: ``code`` int32[n,4], constant table and the Python objects
    behind the table (so that time-dependent Constants can be refreshed every step)."""

    def __init__(self, code, const_sources, aux_functions):
        self.code = code
        self.const_sources = const_sources      # list of Constant | float
        self.aux_functions = aux_functions      # list of Function, index = aux slot

    def constants(self):
        return np.array([float(c.value) if isinstance(c, Constant) else float(c) for c in self.const_sources],
                        dtype=np.float64)


def compile_program(outputs, field_roles, aux_functions=None):
    """outputs: list of 3 expressions (I_ch^k).  field_roles: dict id(Function) -> ('KI', j) |
    ('KE', j) | ('PHIM', 0).  Other Functions get aux slots (shared list ``aux_functions``)."""
    aux_functions = aux_functions if aux_functions is not None else []
    nodes = []                # SSA list: (opname, argidx tuple, imm)
    index = {}                # structural key -> ssa id
    consts = []
    const_index = {}

    def const_slot(src):
        key = ("C", id(src)) if isinstance(src, Constant) else ("L", float(src))
        if key not in const_index:
            const_index[key] = len(consts)
            consts.append(src)
        return const_index[key]

    def emit(key, rec):
        if key in index:
            return index[key]
        nodes.append(rec)
        index[key] = len(nodes) - 1
        return index[key]

    def visit(e):
        e = as_expr(e)
        if isinstance(e, ZeroBaseForm):
            s = const_slot(0.0)
            return emit(("CONST", s), ("CONST", (), s))
        if isinstance(e, Literal):
            s = const_slot(e.value)
            return emit(("CONST", s), ("CONST", (), s))
        if isinstance(e, Constant):
            s = const_slot(e)
            return emit(("CONST", s), ("CONST", (), s))
        if isinstance(e, Coordinate):
            return emit(("X", e.axis), ("X", (), e.axis))
        if isinstance(e, Function):
            role = field_roles.get(id(e))
            if role is None:
                for k, f in enumerate(aux_functions):
                    if f is e:
                        role = ("AUX", k)
                        break
                else:
                    if len(aux_functions) >= KNP_MAX_AUX:
                        raise ValueError(f"more than {KNP_MAX_AUX} auxiliary nodal fields in membrane expressions")
                    aux_functions.append(e)
                    role = ("AUX", len(aux_functions) - 1)
            return emit(role, (role[0], (), role[1]))
        if isinstance(e, Op):
            args = tuple(visit(a) for a in e.args)
            return emit((e.op, args, e.imm), (e.op, args, e.imm))
        raise TypeError(type(e))

    out_ids = [visit(o) for o in outputs]

    # liveness + linear-scan register allocation
    last_use = [-1] * len(nodes)
    for i, (_, args, _) in enumerate(nodes):
        for a in args:
            last_use[a] = i
    for oid in out_ids:
        last_use[oid] = len(nodes) + 1
    free = list(range(KNP_MAX_PROG_REGS - 1, -1, -1))
    reg = [-1] * len(nodes)
    code = []
    for i, (op, args, imm) in enumerate(nodes):
        # registers of operands whose last use is this instruction can be reused for the result,
        # except for SEL whose destination must not alias its operands before the MOV
        releasable = [a for a in set(args) if last_use[a] == i]
        if op != "SEL":
            for a in releasable:
                free.append(reg[a])
        if not free:
            raise ValueError(f"membrane expression needs more than {KNP_MAX_PROG_REGS} registers")
        r = free.pop()
        reg[i] = r
        if op in ("CONST", "KI", "KE", "AUX", "X"):
            code.append((OPS[op], r, imm, 0))
        elif op == "PHIM":
            code.append((OPS[op], r, 0, 0))
        elif op == "POWI":
            code.append((OPS[op], r, reg[args[0]], imm))
        elif op == "SEL":
            c, t, f = args
            code.append((OPS["MOV"], r, reg[f], 0))
            code.append((OPS["SEL"], r, reg[c], reg[t]))
            for a in releasable:
                free.append(reg[a])
        elif len(args) == 1:
            code.append((OPS[op], r, reg[args[0]], 0))
        else:
            code.append((OPS[op], r, reg[args[0]], reg[args[1]]))
        if last_use[i] == -1:          # dead value
            free.append(r)
    for k, oid in enumerate(out_ids):
        code.append((OPS["OUT"], 0, k, reg[oid]))
    return ProgramSpec(np.array(code, dtype=np.int32).reshape(-1, 4), consts, aux_functions)


# ------------------------------------------------------------------------------------------
# host evaluation (setup-time scalars such as the stimulus area; never on the timed path)
# ------------------------------------------------------------------------------------------
def evaluate_numpy(e, env):
    """Evaluate an expression with NumPy. env: {'x': [arrays per axis], 'fields': {id(Function): array}}."""
    e = as_expr(e)
    if isinstance(e, (Literal,)):
        return e.value
    if isinstance(e, ZeroBaseForm):
        return 0.0
    if isinstance(e, Constant):
        return e.value
    if isinstance(e, Coordinate):
        return env["x"][e.axis]
    if isinstance(e, Function):
        return env["fields"][id(e)]
    a = [evaluate_numpy(x, env) for x in e.args]
    op = e.op
    if op == "ADD": return a[0] + a[1]
    if op == "SUB": return a[0] - a[1]
    if op == "MUL": return a[0] * a[1]
    if op == "DIV": return a[0] / a[1]
    if op == "NEG": return -a[0]
    if op == "ABS": return np.abs(a[0])
    if op == "POW": return np.power(a[0], a[1])
    if op == "POWI": return np.power(a[0], e.imm)
    if op == "LN": return np.log(a[0])
    if op == "EXP": return np.exp(a[0])
    if op == "SQRT": return np.sqrt(a[0])
    if op == "MAX": return np.maximum(a[0], a[1])
    if op == "MIN": return np.minimum(a[0], a[1])
    if op == "LT": return (a[0] < a[1]) * 1.0
    if op == "GT": return (a[0] > a[1]) * 1.0
    if op == "LE": return (a[0] <= a[1]) * 1.0
    if op == "GE": return (a[0] >= a[1]) * 1.0
    if op == "EQ": return (a[0] == a[1]) * 1.0
    if op == "AND": return ((np.asarray(a[0]) != 0) & (np.asarray(a[1]) != 0)) * 1.0
    if op == "OR": return ((np.asarray(a[0]) != 0) | (np.asarray(a[1]) != 0)) * 1.0
    if op == "NOT": return (np.asarray(a[0]) == 0) * 1.0
    if op == "SEL": return np.where(np.asarray(a[0]) != 0, a[1], a[2])
    raise ValueError(op)


def interpret_program(spec: ProgramSpec, ki, ke, phim, aux, xq):
    """Pure-Python interpreter of the bytecode (host logic test of the compiler; scalars or arrays)."""
    consts = spec.constants()
    reg = [0.0] * KNP_MAX_PROG_REGS
    out = [0.0, 0.0, 0.0]
    inv = {v: k for k, v in OPS.items()}
    for op, d, a, b in spec.code.tolist():
        name = inv[op]
        if name == "CONST": reg[d] = consts[a]
        elif name == "KI": reg[d] = ki[a]
        elif name == "KE": reg[d] = ke[a]
        elif name == "PHIM": reg[d] = phim
        elif name == "AUX": reg[d] = aux[a]
        elif name == "X": reg[d] = xq[a]
        elif name == "ADD": reg[d] = reg[a] + reg[b]
        elif name == "SUB": reg[d] = reg[a] - reg[b]
        elif name == "MUL": reg[d] = reg[a] * reg[b]
        elif name == "DIV": reg[d] = reg[a] / reg[b]
        elif name == "NEG": reg[d] = -reg[a]
        elif name == "POW": reg[d] = np.power(reg[a], reg[b])
        elif name == "POWI": reg[d] = np.power(reg[a], b)
        elif name == "LN": reg[d] = np.log(reg[a])
        elif name == "EXP": reg[d] = np.exp(reg[a])
        elif name == "SQRT": reg[d] = np.sqrt(reg[a])
        elif name == "MAX": reg[d] = np.maximum(reg[a], reg[b])
        elif name == "MIN": reg[d] = np.minimum(reg[a], reg[b])
        elif name == "ABS": reg[d] = np.abs(reg[a])
        elif name == "LT": reg[d] = (reg[a] < reg[b]) * 1.0
        elif name == "GT": reg[d] = (reg[a] > reg[b]) * 1.0
        elif name == "LE": reg[d] = (reg[a] <= reg[b]) * 1.0
        elif name == "GE": reg[d] = (reg[a] >= reg[b]) * 1.0
        elif name == "EQ": reg[d] = (reg[a] == reg[b]) * 1.0
        elif name == "AND": reg[d] = ((np.asarray(reg[a]) != 0) & (np.asarray(reg[b]) != 0)) * 1.0
        elif name == "OR": reg[d] = ((np.asarray(reg[a]) != 0) | (np.asarray(reg[b]) != 0)) * 1.0
        elif name == "NOT": reg[d] = (np.asarray(reg[a]) == 0) * 1.0
        elif name == "SEL": reg[d] = np.where(np.asarray(reg[a]) != 0, reg[b], reg[d])
        elif name == "MOV": reg[d] = reg[a]
        elif name == "OUT": out[a] = out[a] + reg[b]
        else: raise ValueError(name)
    return out
