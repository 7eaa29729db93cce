"""Multivariate polynomials over the (extension) field -- host-side mirror of the reference's `multivariate.py`
(/root/reference/code/multivariate.py:4-201): `MPolynomial` with the same dictionary representation
(exponent tuple -> coefficient element), constructors (`zero`, `constant`, `variables`, `lift`) and operations
(`+ - * ^`, `evaluate`, `symbolic_degree_bound`, `partial_evaluate`, `degree`, `is_zero`).

The prover does not use this class: constraints live as expression graphs in `air.py`, the GPU evaluates generated code,
and degree bounds come from `air.expand`.  `Table.*_constraints_ext` build MPolynomial objects from those graphs for code
that wants to look at the constraints the way the reference presents them.
"""


class MPolynomial:
    def __init__(self, dictionary):
        self.dictionary = dictionary

    # ---- constructors
    @staticmethod
    def zero():
        return MPolynomial({})

    @staticmethod
    def constant(element):
        return MPolynomial({(0,): element})

    @staticmethod
    def variables(num_variables, field):
        one = field.one()
        return [MPolynomial({tuple(1 if j == i else 0 for j in range(num_variables)): one}) for i in range(num_variables)]

    @staticmethod
    def lift(polynomial, variable_index):
        """univariate polynomial -> the same polynomial in variable `variable_index` of variable_index + 1 variables"""
        if polynomial.is_zero():
            return MPolynomial({})
        field = polynomial.coefficients[0].field
        x = MPolynomial.variables(variable_index + 1, field)[-1]
        acc = MPolynomial({})
        for i, c in enumerate(polynomial.coefficients):
            acc = acc + MPolynomial.constant(c) * (x ^ i)
        return acc

    # ---- helpers
    def _width(self, other=None):
        keys = list(self.dictionary) + (list(other.dictionary) if other is not None else [])
        return max([0] + [len(k) for k in keys])

    @staticmethod
    def _pad(key, width):
        return tuple(key) + (0,) * (width - len(key))

    # ---- ring operations
    def __add__(self, other):
        width = self._width(other)
        out = {self._pad(k, width): v for k, v in self.dictionary.items()}
        for k, v in other.dictionary.items():
            k = self._pad(k, width)
            out[k] = out[k] + v if k in out else v
        return MPolynomial(out)

    def __neg__(self):
        return MPolynomial({k: -v for k, v in self.dictionary.items()})

    def __sub__(self, other):
        return self + (-other)

    def __mul__(self, other):
        width = self._width(other)
        out = {}
        for k0, v0 in self.dictionary.items():
            k0 = self._pad(k0, width)
            for k1, v1 in other.dictionary.items():
                k = tuple(a + b for a, b in zip(k0, self._pad(k1, width)))
                out[k] = out[k] + v0 * v1 if k in out else v0 * v1
        return MPolynomial(out)

    def __xor__(self, exponent):
        if self.is_zero():
            return MPolynomial({})
        field = next(iter(self.dictionary.values())).field
        acc = MPolynomial({(0,) * self._width(): field.one()})
        for bit in bin(exponent)[2:]:
            acc = acc * acc
            if bit == "1":
                acc = acc * self
        return acc

    # ---- queries
    def is_zero(self):
        return all(v.is_zero() for v in self.dictionary.values())

    def degree(self):
        return max((sum(k) for k in self.dictionary), default=-1)

    def evaluate(self, point):
        acc = point[0].field.zero()
        for k, v in self.dictionary.items():
            assert len(point) == len(k), \
                f"number of elements in point {len(point)} does not match with number of variables {len(k)} for polynomial {str(self)}"
            term = v
            for x, e in zip(point, k):
                term = term * (x ^ e)
            acc = acc + term
        return acc

    def symbolic_degree_bound(self, max_degrees):
        """smallest degree bound on the univariate polynomial obtained by substituting polynomials of degrees <= max_degrees
        (all equal, as in the reference: multivariate.py:144-170); -1 for the zero polynomial"""
        if self.degree() == -1:
            return -1
        assert len(max_degrees) >= self._width(), \
            f"max degrees length ({len(max_degrees)}) does not match with number of variables"
        assert max_degrees == [max_degrees[0]] * len(max_degrees), "max degrees must be n repetitions of the same integer"
        bound = -1
        for k, v in self.dictionary.items():
            if not v.is_zero():
                bound = max(bound, sum(e * d for e, d in zip(k, max_degrees)))
        return bound

    def partial_evaluate(self, partial_assignment):
        field = next(iter(self.dictionary.values())).field
        width = self._width()
        substitution = MPolynomial.variables(width, field)
        for index, value in partial_assignment.items():
            substitution[index] = MPolynomial.constant(value)
        out = MPolynomial.zero()
        for k, v in self.dictionary.items():
            term = MPolynomial.constant(v)
            for i, e in enumerate(self._pad(k, width)):
                term = term * (substitution[i] ^ e)
            out = out + term
        return out

    def __str__(self):
        return " + ".join(str(v) + "*" + "*".join("x%d^%d" % (i, e) for i, e in enumerate(k) if e) for k, v in self.dictionary.items())
