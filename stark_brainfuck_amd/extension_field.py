"""The cubic extension F_p[X]/(X^3 - X + 1) -- host-side mirror of the reference's `extension_field.py`
(/root/reference/code/extension_field.py): same classes, attributes (`polynomial`, `field`, `modulus`) and the
same canonical form (the stored polynomial drops trailing zero coefficients, :6-9).  Scalar plumbing only.
"""
from .algebra import BaseField, BaseFieldElement
from .univariate import Polynomial


class ExtensionFieldElement:
    def __init__(self, polynomial, field):
        self.polynomial = Polynomial(polynomial.coefficients[:polynomial.degree() + 1])
        self.field = field

    def __add__(self, right): return self.field.add(self, right)
    def __sub__(self, right): return self.field.subtract(self, right)
    def __mul__(self, right): return self.field.multiply(self, right)
    def __truediv__(self, right): return self.field.divide(self, right)
    def __neg__(self): return self.field.negate(self)
    def inverse(self): return self.field.inverse(self)

    def __xor__(self, exponent):
        acc = self.field.one()
        for bit in bin(exponent)[2:]:
            acc = acc * acc
            if bit == "1":
                acc = acc * self
        return acc

    def __eq__(self, other): return self.polynomial == other.polynomial
    def __neq__(self, other): return not (self.polynomial == other.polynomial)
    def __str__(self): return str(self.polynomial)
    def __repr__(self): return "ExtensionFieldElement(%s)" % self.polynomial
    def is_zero(self): return self.polynomial.is_zero()

    def limbs(self):
        """the three coefficients (c0, c1, c2) as ints, zero padded -- the device representation."""
        c = [e.value for e in self.polynomial.coefficients]
        return c + [0] * (3 - len(c))


class ExtensionField:
    def __init__(self, modulus):
        self.modulus = modulus

    def _is_cubic(self):
        """modulus X^3 - X + 1: products are reduced with X^3 = X - 1, X^4 = X^2 - X on integers instead of Polynomial.__mul__ /
        divide.  Decided on first use (objects read from a proof are rebuilt from their attributes, without __init__)."""
        flag = self.__dict__.get("_cubic")
        if flag is None:
            c = self.modulus.coefficients
            flag = self.__dict__["_cubic"] = len(c) == 4 and [e.value for e in c] == [1, c[0].field.p - 1, 0, 1]
        return flag

    def _base(self):
        return self.modulus.coefficients[0].field

    def zero(self): return ExtensionFieldElement(Polynomial([]), self)
    def one(self): return ExtensionFieldElement(Polynomial([self._base().one()]), self)

    def multiply(self, left, right):
        lc, rc = left.polynomial.coefficients, right.polynomial.coefficients
        if lc and rc and len(lc) <= 3 and len(rc) <= 3 and self._is_cubic():
            # same value, and the same BaseField instance on the result's coefficients, as the generic path below (the product's
            # coefficients are made by the LEFT operand's coefficient field, univariate.py:57-67 / algebra.py:29)
            base = lc[0].field
            p = base.p
            a0, a1, a2 = [e.value for e in lc] + [0] * (3 - len(lc))
            b0, b1, b2 = [e.value for e in rc] + [0] * (3 - len(rc))
            d3, d4 = a1 * b2 + a2 * b1, a2 * b2
            out = [(a0 * b0 - d3) % p, (a0 * b1 + a1 * b0 + d3 - d4) % p, (a0 * b2 + a1 * b1 + a2 * b0 + d4) % p]
            while out and out[-1] == 0:
                out.pop()
            return ExtensionFieldElement(Polynomial([BaseFieldElement(v, base) for v in out]), self)
        return ExtensionFieldElement((left.polynomial * right.polynomial) % self.modulus, self)

    def add(self, left, right): return ExtensionFieldElement(left.polynomial + right.polynomial, self)
    def subtract(self, left, right): return ExtensionFieldElement(left.polynomial - right.polynomial, self)
    def negate(self, operand): return ExtensionFieldElement(-operand.polynomial, self)

    def inverse(self, operand):
        a, b, g = Polynomial.xgcd(operand.polynomial, self.modulus)
        assert a * operand.polynomial + b * self.modulus == g, "bezout relation fails"
        return ExtensionFieldElement(a % self.modulus, self)

    def divide(self, left, right):
        assert not right.is_zero(), "divide by zero"
        a, _, _ = Polynomial.xgcd(right.polynomial, self.modulus)
        return ExtensionFieldElement(left.polynomial * a % self.modulus, self)

    @staticmethod
    def main():
        # X^3 - X + 1 over p = 2^64 - 2^32 + 1, with the same `one` object as constant and leading coefficient
        field = BaseField.main()
        one = BaseFieldElement(1, field)
        return ExtensionField(Polynomial([one, BaseFieldElement(field.p - 1, field), field.zero(), one]))

    def sample(self, byte_array):
        deg = self.modulus.degree()
        chunk = len(byte_array) // deg
        base = self._base()
        return ExtensionFieldElement(Polynomial([base.sample(byte_array[i * chunk:(i + 1) * chunk]) for i in range(deg)]), self)

    def lift(self, base_field_element):
        if type(base_field_element) == ExtensionFieldElement:
            return base_field_element
        return ExtensionFieldElement(Polynomial([base_field_element]), self)

    def from_limbs(self, limbs):
        """element with coefficients (c0, c1, c2) living in this field's own BaseField instance."""
        base = self._base()
        return ExtensionFieldElement(Polynomial([BaseFieldElement(int(v), base) for v in limbs]), self)

    def __call__(self, integer):
        return ExtensionFieldElement(Polynomial([BaseFieldElement(integer, self._base())]), self)
