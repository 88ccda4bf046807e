"""Numeric stand-in for the handful of CasADi names the reference's NLP builder uses
(fixture generation only; build container only; never imported by the product or by tests).

Why: `casadi` is absent from the image, so `ca.SX.sym` / `ca.nlpsol` cannot run and Ipopt cannot be
called.  The reference's builder (`BoundMPC/casadi_ocp_formulation.py:9-391`) and its stage
functions (`bound_mpc_functions.py`) are however plain Python that only *combine* SX objects with
+ - * / ** @, slicing, `.T`, `.reshape`, `vertcat`, `dot`, `sumsqr`, `if_else`, `sin/cos/exp`.  If every
`SX.sym` carries NUMBERS instead of symbols, running the reference's own unmodified code evaluates
its f(x,p) and g(x,p) at those numbers -- which is what fixture G9 records.  This is an evaluation
of the reference's expressions, not an emulation of CasADi's symbolic engine (no graph, no AD).

CasADi matrix semantics reproduced here (every SX is a dense 2-D matrix):
  * `SX.sym(name, n)` is n x 1, `SX.sym(name, n, m)` is n x m; `SX.zeros(n)` is n x 1, `SX.zeros((n, m))` n x m;
  * `x[k]` / `x[a:b]` (ONE index) is column-major linear indexing; the result is a column, or a row when
    x itself is a row (CasADi `Matrix::get_nz`: "if indexed matrix was a row/column vector, make sure that
    the result is too"); `x[r, c]` (TWO indices) is ordinary 2-D indexing and always returns a matrix;
  * assignment `x[idx] = v` accepts v of matching element count in either orientation;
  * `.reshape((-1, 1))` is column-major (this defines the parameter-vector order, casadi_ocp_formulation.py:361-376);
  * elementwise binary operators need equal shapes or one 1 x 1 operand -- anything else RAISES here
    (no numpy-style outer broadcasting, which would silently compute something else);
  * `@` is the matrix product; `dot(a, b)` needs equal shapes and is sum(a*b); comparisons give 0/1;
  * `if_else(c, a, b)` with a 1 x 1 condition picks a branch (numerically identical to CasADi's evaluation).
Values may be complex so that a complex step through the reference's code gives its exact derivatives
(comparisons use real parts, as the `if_else` conditions carry no derivative in CasADi either).
"""
import numpy as np


class SX:
    __array_priority__ = 1000.0     # numpy scalars/arrays defer to the operators below
    __array_ufunc__ = None

    #: called as provider(creation_index, name, n, m) -> (n, m) array of values for a new symbol
    provider = None
    created = []                    # (name, n, m, SX) in creation order of the current build

    def __init__(self, a):
        a = a.a if isinstance(a, SX) else np.asarray(a)
        if a.dtype.kind not in "fc":
            a = a.astype(float)
        if a.ndim == 0:
            a = a.reshape(1, 1)
        elif a.ndim == 1:
            a = a.reshape(-1, 1)
        assert a.ndim == 2
        self.a = np.array(a)        # own copy

    # ---- construction ------------------------------------------------------------------
    @staticmethod
    def sym(name, n=1, m=1):
        if SX.provider is None:
            raise RuntimeError("numeric SX: no value provider installed")
        v = np.asarray(SX.provider(len(SX.created), name, int(n), int(m)))
        s = SX(v.reshape(int(n), int(m)))
        SX.created.append((name, int(n), int(m), s))
        return s

    @staticmethod
    def zeros(*shape):
        if len(shape) == 1 and isinstance(shape[0], (tuple, list)):
            shape = tuple(shape[0])
        if len(shape) == 1:
            shape = (shape[0], 1)
        return SX(np.zeros(shape))

    # ---- shape ---------------------------------------------------------------------------
    @property
    def shape(self):
        return self.a.shape

    @property
    def T(self):
        return SX(self.a.T)

    def reshape(self, shape):
        return SX(np.reshape(self.a, shape, order="F"))

    def numel(self):
        return self.a.size

    def __float__(self):
        assert self.a.size == 1
        return float(np.real(self.a.item()))

    def __repr__(self):
        return "SX(%r)" % (self.a,)

    def __array__(self, dtype=None, copy=None):      # np.array(x), and assignment of a result into a numpy slice (numeric branches)
        a = np.real(self.a) if self.a.dtype.kind == "c" else self.a
        return a.astype(dtype) if dtype is not None else np.array(a)

    # ---- indexing ------------------------------------------------------------------------
    @staticmethod
    def _as_slice(k, n):
        if isinstance(k, slice):
            return k
        k = int(k)
        if k < 0:
            k += n
        if not 0 <= k < n:
            raise IndexError("numeric SX: index %d out of range %d" % (k, n))
        return slice(k, k + 1)

    def __getitem__(self, k):
        if isinstance(k, tuple):
            r, c = k
            return SX(self.a[self._as_slice(r, self.a.shape[0]), self._as_slice(c, self.a.shape[1])])
        flat = self.a.reshape(-1, order="F")[self._as_slice(k, self.a.size)]
        is_row = self.a.shape[0] == 1 and self.a.shape[1] > 1
        return SX(flat.reshape(1, -1) if is_row else flat.reshape(-1, 1))

    def __setitem__(self, k, v):
        v = v.a if isinstance(v, SX) else np.asarray(v)
        if v.dtype.kind == "c" and self.a.dtype.kind != "c":
            self.a = self.a.astype(complex)
        if isinstance(k, tuple):
            r, c = k
            view = self.a[self._as_slice(r, self.a.shape[0]), self._as_slice(c, self.a.shape[1])]
            if v.size == 1:
                view[...] = v.reshape(())
            elif v.shape == view.shape:
                view[...] = v
            elif v.size == view.size and 1 in v.shape and 1 in view.shape:
                view[...] = v.reshape(view.shape)       # vector of the other orientation
            else:
                raise ValueError("numeric SX: assignment size mismatch %r into %r" % (v.shape, view.shape))
            return
        flat = self.a.reshape(-1, order="F")
        sl = self._as_slice(k, self.a.size)
        n = len(range(*sl.indices(self.a.size)))
        if v.size not in (1, n):
            raise ValueError("numeric SX: assignment size mismatch (%d into %d)" % (v.size, n))
        flat[sl] = v.reshape(-1, order="F") if v.size > 1 else v.reshape(())
        self.a = flat.reshape(self.a.shape, order="F")

    # ---- arithmetic -----------------------------------------------------------------------
    @staticmethod
    def _arr(x):
        if isinstance(x, SX):
            return x.a
        x = np.asarray(x)
        if x.ndim == 0:
            return x.reshape(1, 1)
        if x.ndim == 1:
            return x.reshape(-1, 1)
        return x

    @staticmethod
    def _bin(op, x, y):
        a, b = SX._arr(x), SX._arr(y)
        if a.shape != b.shape and a.size != 1 and b.size != 1:
            raise ValueError("numeric SX: dimension mismatch %r vs %r (CasADi would refuse this too)" % (a.shape, b.shape))
        if a.size == 1 and b.size != 1:
            a = a.reshape(())
        elif b.size == 1 and a.size != 1:
            b = b.reshape(())
        return SX(op(a, b))

    def __add__(self, o): return SX._bin(np.add, self, o)
    def __radd__(self, o): return SX._bin(np.add, o, self)
    def __sub__(self, o): return SX._bin(np.subtract, self, o)
    def __rsub__(self, o): return SX._bin(np.subtract, o, self)
    def __mul__(self, o): return SX._bin(np.multiply, self, o)
    def __rmul__(self, o): return SX._bin(np.multiply, o, self)
    def __truediv__(self, o): return SX._bin(np.divide, self, o)
    def __rtruediv__(self, o): return SX._bin(np.divide, o, self)
    def __neg__(self): return SX(-self.a)
    def __pos__(self): return SX(self.a)

    def __pow__(self, e):
        if isinstance(e, SX):
            assert e.a.size == 1
            e = e.a.item()
        if float(e) == int(e):
            return SX(self.a ** int(e))
        return SX(self.a ** e)

    def __matmul__(self, o):
        b = SX._arr(o)
        if self.a.shape[1] != b.shape[0]:
            raise ValueError("numeric SX: matmul dimension mismatch %r @ %r" % (self.a.shape, b.shape))
        return SX(self.a @ b)

    def __rmatmul__(self, o):
        return SX(SX._arr(o)) @ self

    @staticmethod
    def _cmp(op, x, y):
        return SX._bin(lambda a, b: op(np.real(a), np.real(b)).astype(float), x, y)

    def __lt__(self, o): return SX._cmp(np.less, self, o)
    def __le__(self, o): return SX._cmp(np.less_equal, self, o)
    def __gt__(self, o): return SX._cmp(np.greater, self, o)
    def __ge__(self, o): return SX._cmp(np.greater_equal, self, o)
    __hash__ = None


class MX:      # isinstance() placeholder only
    pass


class DM(SX):
    """What CasADi hands back when its functions are called with plain numbers (`ca.if_else(numpy_bool, numpy_row, numpy_row)` in the
    numeric branches of reference_function / error_function, BoundMPC.py:614-752): a dense numeric matrix with the same indexing
    rules as SX; a 1-D numpy array becomes a COLUMN.  `np.array(dm)` gives the 2-D array."""



def _lift(f):
    def g(x):
        return SX(f(x.a)) if isinstance(x, SX) else f(x)
    return g


sin, cos, acos, sqrt, exp = (_lift(f) for f in (np.sin, np.cos, np.arccos, np.sqrt, np.exp))


def _any_sx(args):
    return any(isinstance(x, SX) for x in args)


def dot(a, b):
    if _any_sx((a, b)):
        x, y = SX._arr(a), SX._arr(b)
        if x.shape != y.shape:
            raise ValueError("numeric SX: dot dimension mismatch %r vs %r" % (x.shape, y.shape))
        return SX(np.sum(x * y))
    return float(np.dot(np.asarray(a).ravel(), np.asarray(b).ravel()))


def sumsqr(a):
    if isinstance(a, SX):
        return SX(np.sum(a.a * a.a))          # no conjugate: analytic in the complex step
    return float(np.sum(np.asarray(a) ** 2))


def norm_2(a):
    if isinstance(a, SX):
        return SX(np.sqrt(np.sum(a.a * a.a)))
    return float(np.linalg.norm(a))


def vertcat(*a):
    if _any_sx(a):
        parts = [SX._arr(x) for x in a]
        cols = {p.shape[1] for p in parts if p.size}
        if len(cols) > 1:
            raise ValueError("numeric SX: vertcat column mismatch %r" % ([p.shape for p in parts],))
        return SX(np.vstack([p for p in parts if p.size]))
    return np.concatenate([np.atleast_1d(np.asarray(x, dtype=float)).ravel() for x in a])


def if_else(c, a, b):
    if isinstance(c, SX):
        assert c.a.size == 1, "numeric SX: if_else needs a scalar condition"
        return SX(a) if np.real(c.a.item()) != 0 else SX(b)
    if isinstance(a, np.ndarray) or isinstance(b, np.ndarray) or isinstance(a, SX) or isinstance(b, SX):
        return DM(a) if bool(np.all(c)) else DM(b)      # numeric call: CasADi converts the branches to DM
    return a if c else b


class CapturedNlp:
    """What `ca.nlpsol(name, 'ipopt', prob, opts)` hands back here: the numbers of prob['x','p','f','g']."""

    def __init__(self, prob, opts):
        self.x = SX._arr(prob["x"]).reshape(-1, order="F").copy()
        self.p = SX._arr(prob["p"]).reshape(-1, order="F").copy()
        self.f = SX._arr(prob["f"]).reshape(()).copy()
        self.g = SX._arr(prob["g"]).reshape(-1, order="F").copy()
        self.opts = opts

    def generate_dependencies(self, *a, **k):
        pass


def nlpsol(name, plugin, prob, opts=None):
    assert plugin == "ipopt"
    return CapturedNlp(prob, opts)
