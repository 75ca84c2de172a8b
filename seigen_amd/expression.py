"""`Expression` stand-in: C-string expressions evaluated at node coordinates.

The reference's harnesses pass C code strings to ``firedrake.Expression`` and
interpolate them into DG functions (``tests/eigenmode/eigenmode_2d.py:30-36``,
``tests/explosive_source/explosive_source_lf4.py:36-45``); [upstream] Firedrake
evaluates the string pointwise at the node coordinates ``x[]`` with the keyword
arguments as named constants.  This module parses the same C subset (ternary,
``&& || !``, comparisons, arithmetic, ``pow sin cos exp sqrt ...``, ``pi``)
and evaluates it with numpy over all nodes at once.
"""
import math
import re
import numpy as np

_TOKEN = re.compile(r"\s*(?:(\d+\.\d*(?:[eE][-+]?\d+)?|\.\d+(?:[eE][-+]?\d+)?|\d+(?:[eE][-+]?\d+)?)"
                    r"|([A-Za-z_][A-Za-z_0-9]*)|(&&|\|\||<=|>=|==|!=|[-+*/()<>\[\]?:!,]))")

_FUNCS = {
    "sin": np.sin, "cos": np.cos, "tan": np.tan, "exp": np.exp, "log": np.log, "sqrt": np.sqrt,
    "fabs": np.abs, "abs": np.abs, "pow": np.power, "tanh": np.tanh, "sinh": np.sinh, "cosh": np.cosh,
    "asin": np.arcsin, "acos": np.arccos, "atan": np.arctan, "atan2": np.arctan2,
    "fmin": np.minimum, "fmax": np.maximum, "floor": np.floor, "ceil": np.ceil,
}
_CONSTS = {"pi": math.pi, "M_PI": math.pi, "e": math.e}


def _tokenize(src):
    pos, out = 0, []
    while pos < len(src):
        if src[pos:].strip() == "":
            break
        m = _TOKEN.match(src, pos)
        if not m:
            raise SyntaxError("cannot tokenize expression %r at %d" % (src, pos))
        num, ident, op = m.groups()
        if num is not None:
            out.append(("num", float(num)))
        elif ident is not None:
            out.append(("id", ident))
        else:
            out.append(("op", op))
        pos = m.end()
    out.append(("end", None))
    return out


class _Parser(object):
    """Recursive-descent parser producing nested tuples."""

    def __init__(self, src):
        self.toks = _tokenize(src)
        self.i = 0

    def peek(self):
        return self.toks[self.i]

    def next(self):
        t = self.toks[self.i]
        self.i += 1
        return t

    def accept(self, op):
        t = self.peek()
        if t[0] == "op" and t[1] == op:
            self.i += 1
            return True
        return False

    def expect(self, op):
        if not self.accept(op):
            raise SyntaxError("expected %r near token %d" % (op, self.i))

    def parse(self):
        e = self.ternary()
        if self.peek()[0] != "end":
            raise SyntaxError("trailing tokens in expression")
        return e

    def ternary(self):
        c = self.lor()
        if self.accept("?"):
            a = self.ternary()
            self.expect(":")
            b = self.ternary()
            return ("?:", c, a, b)
        return c

    def lor(self):
        e = self.land()
        while self.accept("||"):
            e = ("||", e, self.land())
        return e

    def land(self):
        e = self.cmp()
        while self.accept("&&"):
            e = ("&&", e, self.cmp())
        return e

    def cmp(self):
        e = self.add()
        while True:
            t = self.peek()
            if t[0] == "op" and t[1] in ("<", ">", "<=", ">=", "==", "!="):
                self.next()
                e = (t[1], e, self.add())
            else:
                return e

    def add(self):
        e = self.mul()
        while True:
            t = self.peek()
            if t[0] == "op" and t[1] in ("+", "-"):
                self.next()
                e = (t[1], e, self.mul())
            else:
                return e

    def mul(self):
        e = self.unary()
        while True:
            t = self.peek()
            if t[0] == "op" and t[1] in ("*", "/"):
                self.next()
                e = (t[1], e, self.unary())
            else:
                return e

    def unary(self):
        if self.accept("-"):
            return ("neg", self.unary())
        if self.accept("+"):
            return self.unary()
        if self.accept("!"):
            return ("not", self.unary())
        return self.primary()

    def primary(self):
        t = self.next()
        if t[0] == "num":
            return ("num", t[1])
        if t[0] == "id":
            if self.accept("("):
                args = []
                if not self.accept(")"):
                    args.append(self.ternary())
                    while self.accept(","):
                        args.append(self.ternary())
                    self.expect(")")
                return ("call", t[1], args)
            if self.accept("["):
                idx = self.ternary()
                self.expect("]")
                return ("index", t[1], idx)
            return ("id", t[1])
        if t[0] == "op" and t[1] == "(":
            e = self.ternary()
            self.expect(")")
            return e
        raise SyntaxError("unexpected token %r" % (t,))


def _truth(v):
    return np.asarray(v) != 0


def _eval(node, env):
    k = node[0]
    if k == "num":
        return node[1]
    if k == "id":
        name = node[1]
        if name in env:
            return env[name]
        if name in _CONSTS:
            return _CONSTS[name]
        raise NameError("unknown identifier %r in Expression" % name)
    if k == "index":
        arr = env[node[1]] if node[1] in env else None
        if arr is None:
            raise NameError("unknown array %r in Expression" % node[1])
        idx = int(_eval(node[2], env))
        return arr[idx]
    if k == "call":
        fn = _FUNCS.get(node[1])
        if fn is None:
            raise NameError("unknown function %r in Expression" % node[1])
        return fn(*[_eval(a, env) for a in node[2]])
    if k == "neg":
        return -_eval(node[1], env)
    if k == "not":
        return np.where(_truth(_eval(node[1], env)), 0.0, 1.0)
    if k == "?:":
        # each branch is evaluated only at the points that take it (a box-limited source is then
        # nearly free outside its box: explosive_source_lf4.py:36-40 on 10^8 nodes)
        c = _truth(_eval(node[1], env))
        if c.ndim == 0:
            return _eval(node[2] if bool(c) else node[3], env)
        out = np.empty(c.shape)
        for mask, branch in ((c, node[2]), (~c, node[3])):
            if mask.any():
                sub = dict(env)
                sub["x"] = [np.broadcast_to(xi, c.shape)[mask] for xi in env["x"]]
                for name, v in env.items():      # array-valued parameters (evaluate_times: `t` per row) follow the points
                    if name != "x" and isinstance(v, np.ndarray) and v.ndim > 0:
                        sub[name] = np.broadcast_to(v, c.shape)[mask]
                out[mask] = _eval(branch, sub)
        return out
    a = _eval(node[1], env)
    b = _eval(node[2], env)
    if k == "+":
        return a + b
    if k == "-":
        return a - b
    if k == "*":
        return a * b
    if k == "/":
        return a / b
    if k == "&&":
        return np.where(_truth(a) & _truth(b), 1.0, 0.0)
    if k == "||":
        return np.where(_truth(a) | _truth(b), 1.0, 0.0)
    ops = {"<": np.less, ">": np.greater, "<=": np.less_equal, ">=": np.greater_equal,
           "==": np.equal, "!=": np.not_equal}
    return np.where(ops[k](a, b), 1.0, 0.0)


def _depends(node, names):
    """Does the expression tree mention one of `names`?"""
    k = node[0]
    if k == "num":
        return False
    if k == "id":
        return node[1] in names
    if k == "index":
        return node[1] in names or _depends(node[2], names)
    if k == "call":
        return any(_depends(a, names) for a in node[2])
    return any(_depends(c, names) for c in node[1:] if isinstance(c, tuple))


def _may_be_nonzero(node, env, unknown, shape):
    """Conservative support of an expression whose parameters `unknown` may take any value:
    bool array of `shape`, True wherever the value can be non-zero for some value of them.
    Sub-expressions that do not mention an unknown are evaluated exactly; a ternary whose
    condition is known selects per point; products intersect, sums unite."""
    if not _depends(node, unknown):
        v = np.asarray(_eval(node, env))
        return np.broadcast_to(v != 0, shape)
    k = node[0]
    if k == "?:":
        a = _may_be_nonzero(node[2], env, unknown, shape)
        b = _may_be_nonzero(node[3], env, unknown, shape)
        if _depends(node[1], unknown):
            # the condition can hold only where it "may be non-zero"; that it may fail is assumed everywhere
            return (_may_be_nonzero(node[1], env, unknown, shape) & a) | b
        c = np.broadcast_to(_truth(_eval(node[1], env)), shape)
        return np.where(c, a, b)
    if k == "neg":
        return _may_be_nonzero(node[1], env, unknown, shape)
    if k == "*":
        return _may_be_nonzero(node[1], env, unknown, shape) & _may_be_nonzero(node[2], env, unknown, shape)
    if k == "/":
        return _may_be_nonzero(node[1], env, unknown, shape)
    if k in ("+", "-"):
        return _may_be_nonzero(node[1], env, unknown, shape) | _may_be_nonzero(node[2], env, unknown, shape)
    if k == "&&":
        return _may_be_nonzero(node[1], env, unknown, shape) & _may_be_nonzero(node[2], env, unknown, shape)
    # identifiers, calls, comparisons, ||, ! of something unknown: anything is possible
    return np.ones(shape, dtype=bool)


def _shape_of(code):
    if isinstance(code, str):
        return ()
    code = tuple(code)
    if len(code) and not isinstance(code[0], str):
        return (len(code), len(tuple(code[0])))
    return (len(code),)


class Expression(object):
    """``Expression(code, **constants)``; ``code`` is a C string, a tuple of
    strings (vector) or a tuple of tuples (tensor).  Constants are attributes
    and may be re-assigned (``expr.t = t`` as ``seigen/elastic.py:287``)."""

    def __init__(self, code=None, **kwargs):
        object.__setattr__(self, "_params", dict(kwargs))
        self.code = code
        self.value_shape = _shape_of(code)
        flat = [code] if isinstance(code, str) else \
            ([c for row in code for c in row] if len(self.value_shape) == 2 else list(code))
        self._asts = [_Parser(str(c)).parse() for c in flat]
        # components with the same source text are evaluated once (a diagonal source tensor has
        # three equal entries and six zeros: explosive_source_lf4.py:36-40)
        self._keys = [str(c).strip() for c in flat]

    def __getattr__(self, name):
        params = object.__getattribute__(self, "_params")
        if name in params:
            return params[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        if name in self._params:
            self._params[name] = value
        else:
            object.__setattr__(self, name, value)

    def user_args(self):
        return dict(self._params)

    def evaluate(self, X):
        """X: [..., dim] coordinates -> [..., *value_shape]."""
        X = np.asarray(X, dtype=np.float64)
        env = dict(self._params)
        env["x"] = [np.ascontiguousarray(X[..., i]) for i in range(X.shape[-1])]
        done = {}
        npts = int(np.prod(X.shape[:-1], dtype=np.int64))
        out = np.empty((len(self._asts), npts))              # component-major: contiguous writes
        for i, (key, ast) in enumerate(zip(self._keys, self._asts)):
            if key not in done:
                done[key] = np.broadcast_to(np.asarray(_eval(ast, env), dtype=np.float64), X.shape[:-1]).reshape(-1)
            out[i] = done[key]
        return np.ascontiguousarray(out.T).reshape(X.shape[:-1] + self.value_shape)

    def evaluate_times(self, X, times, name="t", max_elems=1 << 22):
        """The expression at every (value of the parameter `name`, point): X [npts, dim], times [nt] ->
        [nt, npts, *value_shape].  One vectorised pass (in chunks of rows) instead of one `evaluate` per time:
        the reference re-interpolates a source Expression before every step (elastic.py:285-288), and a 2500-step
        run of explosive_source_lf4.py spent more host time in those 2500 small evaluations than the device
        spent stepping.  Element for element the same numpy operations as `evaluate` with the parameter set."""
        X = np.asarray(X, dtype=np.float64)
        X = X.reshape(-1, X.shape[-1])
        T = np.asarray(times, dtype=np.float64).reshape(-1)
        npts, nt, ncomp = X.shape[0], len(T), len(self._asts)
        out = np.empty((nt, npts, ncomp))
        rows = max(1, int(max_elems) // max(npts, 1))
        for k0 in range(0, nt, rows):
            k1 = min(nt, k0 + rows)
            shape = (k1 - k0, npts)
            env = dict(self._params)
            env["x"] = [np.broadcast_to(X[None, :, i], shape) for i in range(X.shape[1])]
            env[name] = np.broadcast_to(T[k0:k1, None], shape)
            done = {}
            for i, (key, ast) in enumerate(zip(self._keys, self._asts)):
                if key not in done:
                    done[key] = np.broadcast_to(np.asarray(_eval(ast, env), dtype=np.float64), shape)
                out[k0:k1, :, i] = done[key]
        return out.reshape((nt, npts) + self.value_shape)

    def nonzero_mask(self, X):
        """X: [..., dim] -> bool [...]: some component is non-zero there (no value tensor is built)."""
        X = np.asarray(X, dtype=np.float64)
        env = dict(self._params)
        env["x"] = [np.ascontiguousarray(X[..., i]) for i in range(X.shape[-1])]
        mask = np.zeros(X.shape[:-1], dtype=bool)
        for key in set(self._keys):
            v = np.asarray(_eval(self._asts[self._keys.index(key)], env))
            if v.ndim == 0:
                if v != 0:
                    mask[...] = True
            else:
                mask |= np.broadcast_to(v, mask.shape) != 0
        return mask

    def support_mask(self, X, unknown=("t",)):
        """X: [..., dim] -> bool [...]: True wherever some component can be non-zero for SOME value of
        the parameters `unknown` (a superset of the support at every instant; exact for the usual
        `spatial condition ? f(t) : 0` sources, explosive_source_lf4.py:36-40)."""
        X = np.asarray(X, dtype=np.float64)
        env = {k: v for k, v in self._params.items() if k not in unknown}
        env["x"] = [np.ascontiguousarray(X[..., i]) for i in range(X.shape[-1])]
        mask = np.zeros(X.shape[:-1], dtype=bool)
        for key in set(self._keys):
            mask |= _may_be_nonzero(self._asts[self._keys.index(key)], env, set(unknown), mask.shape)
        return mask
