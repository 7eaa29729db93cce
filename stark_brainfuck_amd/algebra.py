"""Scalar object model of the base field F_p, p = 2^64 - 2^32 + 1 -- the host-side mirror of the reference's
`algebra.py` (/root/reference/code/algebra.py).  Same names, arguments, attributes (`value`, `field`, `p`) and
error behaviour, so code written against the reference runs unchanged.  These objects are plumbing: bulk
arithmetic never loops over them -- vectors go to HBM (arrays.py) and through the HIP kernels.
"""

P_GOLDILOCKS = (1 << 64) - (1 << 32) + 1
_ROOT_2_32 = 1753635133440165772          # algebra.py:126-129: 7^(2^32-1 cofactor), order 2^32


def xgcd(x, y):
    """extended Euclid on integers: returns (a, b, g) with a*x + b*y == g   (algebra.py:1-12)."""
    r0, r1, s0, s1, t0, t1 = x, y, 1, 0, 0, 1
    while r1:
        q = r0 // r1
        r0, r1, s0, s1, t0, t1 = r1, r0 - q * r1, s1, s0 - q * s1, t1, t0 - q * t1
    return s0, t0, r0


class BaseFieldElement:
    """algebra.py:15-73"""

    def __init__(self, value, field):
        self.value = value
        self.field = field

    def __add__(self, right): return self.field.add(self, right)
    def __sub__(self, right): return self.field.subtract(self, right)
    def __mul__(self, right): return self.field.multiply(self, right)
    def __truediv__(self, right): return self.field.divide(self, right)
    def __neg__(self): return self.field.negate(self)
    def inverse(self): return self.field.inverse(self)

    def __xor__(self, exponent):
        # exponentiation, spelled `^` in the reference (algebra.py:39-46)
        return BaseFieldElement(pow(self.value, exponent, self.field.p) if exponent >= 0 else
                                pow(self.inverse().value, -exponent, self.field.p), self.field)

    def __eq__(self, other): return self.value == other.value
    def __neq__(self, other): return self.value != other.value
    def __hash__(self): return self.value
    def __str__(self): return str(self.value)
    def __repr__(self): return "BaseFieldElement(%d)" % self.value
    def __bytes__(self): return str(self.value).encode()
    def is_zero(self): return self.value == 0

    def has_order_po2(self, order):
        assert order & (order - 1) == 0
        if self.value == 1 and order == 1:
            return True
        return (self ^ order).value == 1 and (self ^ (order // 2)).value != 1


class BaseField:
    """algebra.py:76-145"""

    def __init__(self, p):
        self.p = p

    def lift(self, bfe): return bfe
    def zero(self): return BaseFieldElement(0, self)
    def one(self): return BaseFieldElement(1, self)
    def add(self, left, right): return BaseFieldElement((left.value + right.value) % self.p, self)
    def subtract(self, left, right): return BaseFieldElement((left.value - right.value) % self.p, self)
    def multiply(self, left, right): return BaseFieldElement(left.value * right.value % self.p, self)
    def negate(self, operand): return BaseFieldElement(-operand.value % self.p, self)

    def inverse(self, operand):
        a, _, _ = xgcd(operand.value, self.p)
        return BaseFieldElement(a % self.p, self)

    def divide(self, left, right):
        assert not right.is_zero(), "divide by zero"
        a, _, _ = xgcd(right.value, self.p)
        return BaseFieldElement(left.value * a % self.p, self)

    @staticmethod
    def main():
        return BaseField(P_GOLDILOCKS)

    def generator(self):
        assert self.p == P_GOLDILOCKS, "Do not know generator for other fields beyond 2^64 - 2^32 + 1"
        return BaseFieldElement(7, self)

    def primitive_nth_root(self, n):
        assert self.p == P_GOLDILOCKS, "Unknown field, can't return root of unity."
        assert n <= 1 << 32 and (n & (n - 1)) == 0, \
            "Field does not have nth root of unity where n > 2^32 or not power of two."
        return BaseFieldElement(pow(_ROOT_2_32, (1 << 32) // n, self.p), self)

    def sample(self, byte_array):
        return BaseFieldElement(int.from_bytes(bytes(byte_array), "big") % self.p, self)

    def __call__(self, integer):
        return BaseFieldElement(integer % self.p, self)
