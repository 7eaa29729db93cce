"""The arithmetic intermediate representation of the Brainfuck VM, stated ONCE as expression graphs and interpreted
three ways:

  * `expand()`            exact multivariate expansion -> degree bounds, as the reference obtains them from its symbolic
                          MPolynomial objects (/root/reference/code/multivariate.py:144-170, table.py:170-173,238-247,300-304);
  * `evaluate()`          numeric evaluation at one point on the host (the verifier, and the CPU side of the parity tests);
  * tools/gen_air.py      straight-line HIP code for the quotient kernels (csrc/air_generated.hpp).

The constraint sets restate the reference's tables:
  processor    /root/reference/code/processor_table.py:51-160 (base AIR), :211-327 (extension, boundary, terminal)
  instruction  instruction_table.py:27-48, :75-165
  memory       memory_table.py:45-96, :118-170
  input/output io_table.py:32-75
Columns of a table are numbered base columns first, then extension columns; variable (c, False) is column c of the current
row, (c, True) of the next row.  Challenges are numbered a b c d e f alpha beta gamma delta eta = 0..10, terminals as
BrainfuckStark.get_terminals (brainfuck_stark.py:103-109).
"""
P = (1 << 64) - (1 << 32) + 1

# ---------------------------------------------------------------------------------------------------------------
# extension-field arithmetic on int triples (c0, c1, c2), X^3 = X - 1   (extension_field.py:65-86)


def xadd(a, b): return ((a[0] + b[0]) % P, (a[1] + b[1]) % P, (a[2] + b[2]) % P)
def xsub(a, b): return ((a[0] - b[0]) % P, (a[1] - b[1]) % P, (a[2] - b[2]) % P)
def xneg(a): return (-a[0] % P, -a[1] % P, -a[2] % P)
def xlift(v): return (v % P, 0, 0)


X0, X1 = (0, 0, 0), (1, 0, 0)


def xmul(a, b):
    a0, a1, a2 = a
    b0, b1, b2 = b
    d0 = a0 * b0
    d1 = a0 * b1 + a1 * b0
    d2 = a0 * b2 + a1 * b1 + a2 * b0
    d3 = a1 * b2 + a2 * b1
    d4 = a2 * b2
    return ((d0 - d3) % P, (d1 + d3 - d4) % P, (d2 + d4) % P)


def xscale(a, s): return (a[0] * s % P, a[1] * s % P, a[2] * s % P)


def xpow(a, e):
    acc = X1
    for bit in bin(e)[2:]:
        acc = xmul(acc, acc)
        if bit == "1":
            acc = xmul(acc, a)
    return acc


def xinv(a):
    """inverse through the 3x3 multiplication matrix (columns a, a*X, a*X^2)."""
    cols = [a, xmul(a, (0, 1, 0)), xmul(a, (0, 0, 1))]
    m = [[cols[j][i] for j in range(3)] for i in range(3)]
    c00 = (m[1][1] * m[2][2] - m[1][2] * m[2][1]) % P
    c01 = (m[1][2] * m[2][0] - m[1][0] * m[2][2]) % P
    c02 = (m[1][0] * m[2][1] - m[1][1] * m[2][0]) % P
    det = (m[0][0] * c00 + m[0][1] * c01 + m[0][2] * c02) % P
    di = pow(det, P - 2, P)
    return (c00 * di % P, c01 * di % P, c02 * di % P)


# ---------------------------------------------------------------------------------------------------------------
# expression graphs

class E:
    """node of a constraint expression.  op: 'v' variable (val = (column, next_row)), 'k' integer constant,
    'c' challenge, 't' terminal, 'p' table parameter, '+', '-', '*', 'n' (negation)."""
    __slots__ = ("op", "a", "b", "val")

    def __init__(self, op, a=None, b=None, val=None):
        self.op, self.a, self.b, self.val = op, a, b, val

    @staticmethod
    def wrap(x):
        return x if isinstance(x, E) else E("k", val=int(x) % P)

    def __add__(self, o): return E("+", self, E.wrap(o))
    def __radd__(self, o): return E("+", E.wrap(o), self)
    def __sub__(self, o): return E("-", self, E.wrap(o))
    def __rsub__(self, o): return E("-", E.wrap(o), self)
    def __mul__(self, o): return E("*", self, E.wrap(o))
    def __rmul__(self, o): return E("*", E.wrap(o), self)
    def __neg__(self): return E("n", self)


def var(col, nxt=False): return E("v", val=(col, bool(nxt)))
def const(v): return E("k", val=int(v) % P)
def chal(i): return E("c", val=i)
def term(i): return E("t", val=i)
def param(i): return E("p", val=i)


A, B, C, D, EE, F, ALPHA, BETA, GAMMA, DELTA, ETA = range(11)
INSTRUCTIONS = "[]<>,.+-"        # order of processor_table.py:44


def _prod(factors):
    acc = const(1)
    for f in factors:
        acc = acc * f
    return acc


def deselector(instr, x):
    """zero on every instruction except `instr`   (processor_table.py:39-49)"""
    return _prod(x - ord(c) for c in INSTRUCTIONS if c != instr)


def instruction_zerofier(x):
    """zero on all eight instructions   (processor_table.py:201-208)"""
    return _prod(x - ord(c) for c in "[]<>+-,.")


class TableAir:
    """constraint set of one table: lists of expressions over 2*full_width variables."""
    name = ""
    base_width = full_width = 0
    num_params = 0

    def boundary(self): raise NotImplementedError
    def transition(self): raise NotImplementedError
    def terminal(self): raise NotImplementedError

    def all(self):
        """the three constraint lists; built once (the graphs are symbolic in challenges, terminals and parameters)"""
        if getattr(self, "_all", None) is None:
            self._all = [("boundary", self.boundary()), ("transition", self.transition()), ("terminal", self.terminal())]
        return self._all


class ProcessorAir(TableAir):
    name, base_width, full_width = "processor", 7, 11
    CLK, IP, CI, NI, MP, MV, MVI, IPP, MPP, IEV, OEV = range(11)

    def _instruction_polynomials(self, instr, cur, nxt):
        """three polynomials per instruction: instruction pointer, memory pointer, memory value (processor_table.py:51-121)"""
        ip, ni, mp, mv, mvi = cur[self.IP], cur[self.NI], cur[self.MP], cur[self.MV], cur[self.MVI]
        ip_n, mp_n, mv_n = nxt[self.IP], nxt[self.MP], nxt[self.MV]
        mv_is_zero = mv * mvi - 1
        zero = None
        if instr == "[":
            p = [mv * (ip_n - ip - 2) + mv_is_zero * (ip_n - ni), mp_n - mp, mv_n - mv]
        elif instr == "]":
            p = [mv_is_zero * (ip_n - ip - 2) + mv * (ip_n - ni), mp_n - mp, mv_n - mv]
        elif instr == "<":
            p = [ip_n - ip - 1, mp_n - mp + 1, zero]
        elif instr == ">":
            p = [ip_n - ip - 1, mp_n - mp - 1, zero]
        elif instr == "+":
            p = [ip_n - ip - 1, mp_n - mp, mv_n - mv - 1]
        elif instr == "-":
            p = [ip_n - ip - 1, mp_n - mp, mv_n - mv + 1]
        elif instr == ",":
            p = [ip_n - ip - 1, mp_n - mp, zero]
        else:  # "."
            p = [ip_n - ip - 1, mp_n - mp, mv_n - mv]
        return p

    def transition(self):
        cur = [var(i) for i in range(11)]
        nxt = [var(i, True) for i in range(11)]
        ci, mv, mvi = cur[self.CI], cur[self.MV], cur[self.MVI]
        polys = [None, None, None]
        for c in "[]<>+-,.":                                  # processor_table.py:131
            instr = self._instruction_polynomials(c, cur, nxt)
            des = deselector(c, ci)
            for i in range(3):
                if instr[i] is None:
                    continue
                t = des * (instr[i] * ci)                     # instruction polynomials vanish on padding rows (ci = 0)
                polys[i] = t if polys[i] is None else polys[i] + t
        mv_is_zero = mv * mvi - 1
        polys += [nxt[self.CLK] - cur[self.CLK] - 1, mv * mv_is_zero, mvi * mv_is_zero]
        a, b, c, d, e, f = (chal(i) for i in (A, B, C, D, EE, F))
        ipp, mpp, iev, oev = cur[self.IPP], cur[self.MPP], cur[self.IEV], cur[self.OEV]
        polys.append((ipp * (chal(ALPHA) - a * cur[self.IP] - b * ci - c * cur[self.NI]) - nxt[self.IPP]) * ci
                     + instruction_zerofier(ci) * (ipp - nxt[self.IPP]))
        polys.append((mpp * (chal(BETA) - d * cur[self.CLK] - e * cur[self.MP] - f * mv) - nxt[self.MPP]) * ci
                     + (mpp - nxt[self.MPP]) * instruction_zerofier(ci))
        polys.append((nxt[self.IEV] - iev * chal(GAMMA) - nxt[self.MV]) * deselector(",", ci) * ci
                     + (nxt[self.IEV] - iev) * (ord(",") - ci))
        polys.append((nxt[self.OEV] - oev * chal(DELTA) - mv) * deselector(".", ci) * ci
                     + (nxt[self.OEV] - oev) * (ord(".") - ci))
        return polys

    def boundary(self):
        return [var(i) for i in (self.CLK, self.IP, self.MP, self.MV, self.MVI, self.IEV, self.OEV)]

    def terminal(self):
        x = [var(i) for i in range(11)]
        d, e, f = chal(D), chal(EE), chal(F)
        ci = x[self.CI]
        return [term(0) - x[self.IPP],
                (term(1) - x[self.MPP] * (chal(BETA) - d * x[self.CLK] - e * x[self.MP] - f * x[self.MV])) * ci
                + (term(1) - x[self.MPP]) * instruction_zerofier(ci),
                term(2) - x[self.IEV],
                term(3) - x[self.OEV]]


class InstructionAir(TableAir):
    name, base_width, full_width = "instruction", 3, 5
    ADDR, CI, NI, PERM, EVAL = range(5)

    def transition(self):
        addr, ci, ni, perm, ev = (var(i) for i in range(5))
        addr_n, ci_n, ni_n, perm_n, ev_n = (var(i, True) for i in range(5))
        a, b, c = chal(A), chal(B), chal(C)
        polys = [(addr_n - addr - 1) * (addr_n - addr),
                 (addr_n - addr) * (ni - ci_n),
                 (addr_n - addr - 1) * (ci_n - ci),
                 (addr_n - addr - 1) * (ni_n - ni)]
        polys.append((perm * (chal(ALPHA) - a * addr_n - b * ci_n - c * ni_n) - perm_n) * ci * (addr + 1 - addr_n)
                     + instruction_zerofier(ci) * (perm - perm_n)
                     + (addr - addr_n) * (perm - perm_n))
        polys.append((addr_n - addr) * (ev * chal(ETA) + a * addr_n + b * ci_n + c * ni_n - ev_n)
                     + (addr_n - addr - 1) * (ev - ev_n))
        return polys

    def boundary(self):
        x = [var(i) for i in range(5)]
        return [x[self.ADDR], x[self.EVAL] - chal(A) * x[self.ADDR] - chal(B) * x[self.CI] - chal(C) * x[self.NI]]

    def terminal(self):
        return [var(self.PERM) - term(0), var(self.EVAL) - term(4)]


class MemoryAir(TableAir):
    name, base_width, full_width = "memory", 4, 5
    CLK, MP, MV, DUMMY, PERM = range(5)

    def transition(self):
        clk, mp, mv, dm, perm = (var(i) for i in range(5))
        clk_n, mp_n, mv_n, dm_n, perm_n = (var(i, True) for i in range(5))
        polys = [(mp_n - mp - 1) * (mp_n - mp),
                 (mp_n - mp) * mv_n,
                 (dm_n - 1) * dm_n,
                 dm * (mp_n - mp),
                 dm * (mv_n - mv),
                 (mp_n - 1 - mp) * (clk_n - 1 - clk)]
        polys.append((perm * (chal(BETA) - chal(D) * clk - chal(EE) * mp - chal(F) * mv) - perm_n) * (1 - dm)
                     + (perm - perm_n) * dm)
        return polys

    def boundary(self):
        return [var(self.CLK), var(self.MP), var(self.MV)]

    def terminal(self):
        clk, mp, mv, dm, perm = (var(i) for i in range(5))
        return [(perm * (chal(BETA) - chal(D) * clk - chal(EE) * mp - chal(F) * mv) - term(1)) * (1 - dm)
                + (perm - term(1)) * dm]


class IOAir(TableAir):
    """input (challenge gamma = 8, terminal 2) or output (delta = 9, terminal 3) table; parameter 0 is
    iota^(height - length), the factor the running evaluation picks up over the padding rows (io_table.py:54-75)."""
    base_width, full_width, num_params = 1, 2, 1
    COL, EVAL = 0, 1

    def __init__(self, name, challenge_index, terminal_index):
        self.name, self.challenge_index, self.terminal_index = name, challenge_index, terminal_index

    def transition(self):
        return [var(self.EVAL) * chal(self.challenge_index) + var(self.COL, True) - var(self.EVAL, True)]

    def boundary(self):
        return [var(self.EVAL) - var(self.COL)]

    def terminal(self):
        return [var(self.EVAL) - term(self.terminal_index) * param(0)]


TABLE_AIRS = [ProcessorAir(), InstructionAir(), MemoryAir(), IOAir("input", GAMMA, 2), IOAir("output", DELTA, 3)]

# ---------------------------------------------------------------------------------------------------------------
# interpretation 1: exact expansion (dictionary exponent vector -> extension coefficient)


_EXP_BITS = 6          # bits per variable in a packed exponent vector (degrees stay below 64: the largest constraint has 11)


def expand(e, nvars, challenges, terminals, params=(), _memo=None):
    """multivariate expansion of expression e as {packed exponent vector: (c0, c1, c2)}: the exponent of variable i sits in
    bits [6 i, 6 i + 6) of the key, so multiplying monomials is adding keys; variable (c, next) has index c + next * (nvars // 2).
    Zero coefficients are dropped.  (`unpack_exponents` turns a key back into the reference's exponent tuple.)"""
    memo = {} if _memo is None else _memo
    key = id(e)
    if key in memo:
        return memo[key]
    op = e.op
    if op == "v":
        col, nxt = e.val
        idx = col + (nvars // 2 if nxt else 0)
        r = {1 << (_EXP_BITS * idx): X1}
    elif op == "k":
        r = {0: xlift(e.val)} if e.val % P else {}
    elif op in "ctp":
        v = {"c": challenges, "t": terminals, "p": params}[op][e.val]
        r = {0: tuple(v)} if any(v) else {}
    elif op == "n":
        r = {k: xneg(v) for k, v in expand(e.a, nvars, challenges, terminals, params, memo).items()}
    else:
        x = expand(e.a, nvars, challenges, terminals, params, memo)
        y = expand(e.b, nvars, challenges, terminals, params, memo)
        if op in "+-":
            r = dict(x)
            for k, v in y.items():
                w = xadd(r.get(k, X0), v) if op == "+" else xsub(r.get(k, X0), v)
                if any(w):
                    r[k] = w
                else:
                    r.pop(k, None)
        else:
            r = {}
            get = r.get
            for k0, v0 in x.items():
                for k1, v1 in y.items():
                    k = k0 + k1
                    w = xadd(get(k, X0), xmul(v0, v1))
                    if any(w):
                        r[k] = w
                    else:
                        r.pop(k, None)
    memo[key] = r
    return r


def unpack_exponents(key, nvars):
    return tuple((key >> (_EXP_BITS * i)) & ((1 << _EXP_BITS) - 1) for i in range(nvars))


def total_degrees(expansion):
    """the distinct total degrees of the (non-zero) monomials of an expansion, as a sorted tuple"""
    mask = (1 << _EXP_BITS) - 1
    out = set()
    for key in expansion:
        total = 0
        while key:
            total += key & mask
            key >>= _EXP_BITS
        out.add(total)
    return tuple(sorted(out))


def symbolic_degree_bound(expansion, max_degree):
    """multivariate.py:144-170 with a uniform bound on every argument: max over the non-zero monomials of
    (total degree * max_degree); -1 for the zero polynomial."""
    return max([-1] + [t * max_degree for t in total_degrees(expansion)])


# ---------------------------------------------------------------------------------------------------------------
# interpretation 2: numeric evaluation at one point (extension values throughout)


def evaluate(e, cur, nxt, challenges, terminals, params=(), _memo=None):
    memo = {} if _memo is None else _memo
    key = id(e)
    if key in memo:
        return memo[key]
    op = e.op
    if op == "v":
        col, n = e.val
        r = (nxt if n else cur)[col]
    elif op == "k":
        r = xlift(e.val)
    elif op in "ctp":
        r = tuple({"c": challenges, "t": terminals, "p": params}[op][e.val])
    elif op == "n":
        r = xneg(evaluate(e.a, cur, nxt, challenges, terminals, params, memo))
    else:
        x = evaluate(e.a, cur, nxt, challenges, terminals, params, memo)
        y = evaluate(e.b, cur, nxt, challenges, terminals, params, memo)
        r = xadd(x, y) if op == "+" else (xsub(x, y) if op == "-" else xmul(x, y))
    memo[key] = r
    return r
