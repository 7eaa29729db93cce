"""The reference's own property tests for the scalar layer, restated against this package's host-side mirrors (CPU only).

Each test names the reference test it follows.  The bulk counterparts of test_ntt.py / test_merkle.py / test_fri.py /
test_bfs.py run on the GPU (tests/test_gpu_parity.py, tests/test_gpu_stark.py).
"""
import os

import pytest

import stark_brainfuck_amd as sb
from stark_brainfuck_amd.multivariate import MPolynomial
from stark_brainfuck_amd.vm import VirtualMachine

HELLO = "++++++++[>++++[>++>+++>+++>+<<<<-]>+>+>->>+[<]<-]>>.>---.+++++++..+++.>>.<-.<.+++.------.--------.>>+.>++."


def test_extension_inverse():
    """/root/reference/code/test_extension_field.py:5-8"""
    field = sb.ExtensionField.main()
    for _ in range(20):
        a = field.sample(os.urandom(8 * 3))
        if a.is_zero():
            continue
        assert a * a.inverse() == field.one()


def test_polynomial_xgcd():
    """/root/reference/code/test_extension_field.py:11-21: Bezout relation and inverse modulo a coprime polynomial"""
    field = sb.BaseField.main()
    x = sb.Polynomial([field.sample(os.urandom(8)) for _ in range(10)])
    y = sb.Polynomial([field.sample(os.urandom(8)) for _ in range(13)])
    a, b, g = sb.Polynomial.xgcd(x, y)
    assert a * x + y * b == g
    assert (a * x) % y == sb.Polynomial([field.one()])


def test_symbolic_bounds_zero_coefficient_zero_input():
    """/root/reference/code/test_multivariate.py:9-23"""
    field = sb.ExtensionField.main()
    assert MPolynomial({(0, 1): field.one()}).symbolic_degree_bound([-1, -1]) == -1
    assert MPolynomial({(0, 1): field.one(), (0, 0): field.zero()}).symbolic_degree_bound([-1, -1]) == -1
    assert MPolynomial({(0, 1): field.one(), (100, 42): field.zero()}).symbolic_degree_bound([-1, -1]) == -1


def test_symbolic_bounds_zero_coefficient_non_zero_input():
    """/root/reference/code/test_multivariate.py:26-44"""
    field = sb.ExtensionField.main()
    degrees = [3, 3, 3]
    assert MPolynomial({(0, 2, 1): field.one(), (0, 0, 1): field.one()}).symbolic_degree_bound(degrees) == 9
    assert MPolynomial({(0, 2, 1): field.one(), (0, 0, 1): field.one(), (3, 3, 4): field.zero()}).symbolic_degree_bound(degrees) == 9
    assert MPolynomial({(0, 2, 1): field.one(), (0, 0, 1): field.one(), (3, 3, 4): field.one()}).symbolic_degree_bound(degrees) != 9


def test_multivariate_ring_laws_and_evaluation():
    """evaluate is a ring homomorphism (what table.py relies on when it evaluates constraint polynomials on rows), and
    the symbolic bound dominates the degree of the composed univariate polynomial (multivariate.py:144-170)"""
    field = sb.BaseField.main()
    x, y, z = MPolynomial.variables(3, field)
    c = MPolynomial.constant(field(7))
    f = x * y + c * z - (y ^ 3)
    g = (x + z) * (x - z) + c
    point = [field.sample(os.urandom(8)) for _ in range(3)]
    assert (f * g).evaluate(point) == f.evaluate(point) * g.evaluate(point)
    assert (f + g).evaluate(point) == f.evaluate(point) + g.evaluate(point)
    assert (f - f).is_zero()
    # substitute univariate polynomials of degree <= 4 and compare degrees
    polys = [sb.Polynomial([field.sample(os.urandom(8)) for _ in range(5)]) for _ in range(3)]
    composed = sb.Polynomial([])
    for k, v in (f * g).dictionary.items():
        term = sb.Polynomial([v])
        for p, e in zip(polys, k):
            term = term * (p ^ e)
        composed = composed + term
    assert composed.degree() <= (f * g).symbolic_degree_bound([4, 4, 4])
    assert MPolynomial.lift(polys[0], 2).evaluate(point) == polys[0].evaluate(point[2])
    # partial evaluation then full evaluation agrees with direct evaluation
    partial = f.partial_evaluate({1: point[1]})
    assert partial.evaluate(point) == f.evaluate(point)


def test_univariate_properties():
    """the univariate layer the tutorial checks by use: distributivity, division with remainder, interpolation through a
    domain, the zerofier of a domain, colinearity (/root/reference/code/univariate.py:70-170)"""
    field = sb.BaseField.main()
    rnd = lambda n: sb.Polynomial([field.sample(os.urandom(8)) for _ in range(n)])
    a, b, c = rnd(6), rnd(9), rnd(4)
    assert a * (b + c) == a * b + a * c
    q, r = sb.Polynomial.divide(a * b + c, b)
    assert q == a and r == c
    domain = [field(i) for i in range(1, 9)]
    values = [field.sample(os.urandom(8)) for _ in domain]
    p = sb.Polynomial.interpolate_domain(domain, values)
    assert p.degree() < len(domain) and p.evaluate_domain(domain) == values
    z = sb.Polynomial.zerofier_domain(domain)
    assert z.degree() == len(domain) and all(z.evaluate(d).is_zero() for d in domain)
    line = rnd(2)
    points = [(x, line.evaluate(x)) for x in (field(3), field(11), field(500))]
    from stark_brainfuck_amd import univariate
    assert sb.colinear(points) and univariate.test_colinearity(points) and not hasattr(sb, "test_colinearity")
    points[1] = (points[1][0], points[1][1] + field.one())
    assert not sb.colinear(points)
    offset = field(5)
    assert a.scale(offset).evaluate(field(9)) == a.evaluate(offset * field(9))


def test_vm_hello_world():
    """/root/reference/code/test_vm.py:6-15"""
    running_time, input_data, output_data = VirtualMachine.execute(HELLO)
    assert "".join(output_data) == "Hello World!\n"
    program = VirtualMachine.compile(HELLO)
    assert VirtualMachine.run(program)[0] == running_time


def test_vm_states():
    """/root/reference/code/test_vm.py:18-23: a program whose loop is skipped; five tables come back, one processor
    row per cycle"""
    program = VirtualMachine.compile(">>[++-]<")
    running_time, _, _ = VirtualMachine.run(program)
    processor, memory, instruction, inp, out = VirtualMachine.simulate(program)      # vm.py:306
    assert len(processor) == running_time and len(memory) == running_time - 1    # the final row (instruction 0) is left out, no dummy rows needed here (memory_table.py:21-38)
    assert len(instruction) == running_time + len(program)
    assert len(inp) == 0 and len(out) == 0


def test_vm_input_output_tables():
    program = VirtualMachine.compile(",+.,.")
    running_time, inputs, outputs = VirtualMachine.run(program, input_data=list("ax"))
    assert outputs == ["b", "x"]
    matrices = VirtualMachine.simulate(program, input_data=list("ax"))
    assert [int(r[0].value) for r in matrices[3]] == [ord("a"), ord("x")]
    assert [int(r[0].value) for r in matrices[4]] == [ord("b"), ord("x")]


def test_cubic_multiplication_shortcut_equals_the_generic_reduction():
    """ExtensionField.multiply reduces with X^3 = X - 1 on integers; value, canonical form and the BaseField instance of the
    result's coefficients must be those of (left.polynomial * right.polynomial) % modulus (extension_field.py:65-67)"""
    import random
    from stark_brainfuck_amd.algebra import BaseField, BaseFieldElement
    from stark_brainfuck_amd.extension_field import ExtensionFieldElement
    from stark_brainfuck_amd.univariate import Polynomial
    xf = sb.ExtensionField.main()
    other = BaseField(xf.modulus.coefficients[0].field.p)           # a second instance, as the VM's field is
    rng = random.Random(5)
    for trial in range(500):
        a = xf.from_limbs([rng.randrange(0, other.p) for _ in range(rng.randrange(0, 4))])
        b = ExtensionFieldElement(Polynomial([BaseFieldElement(rng.randrange(0, other.p), other) for _ in range(rng.randrange(0, 4))]), xf)
        for left, right in ((a, b), (b, a)):
            fast = left * right
            slow = ExtensionFieldElement((left.polynomial * right.polynomial) % xf.modulus, xf)
            assert fast == slow
            assert [c.value for c in fast.polynomial.coefficients] == [c.value for c in slow.polynomial.coefficients]
            assert [c.field is other for c in fast.polynomial.coefficients] == [c.field is other for c in slow.polynomial.coefficients]


def test_three_point_colinearity_shortcut_equals_interpolation():
    """colinear() on three points uses cross products; the verdict must be that of the reference's definition -- the Lagrange
    interpolant has degree exactly 1 (univariate.py:190-194) -- for lines, constants, zero and generic points"""
    import random
    from stark_brainfuck_amd.univariate import Polynomial
    xf, f = sb.ExtensionField.main(), sb.BaseField.main()
    rng = random.Random(9)
    rnd = lambda: xf.from_limbs([rng.randrange(0, f.p) for _ in range(3)])
    slow = lambda pts: Polynomial.interpolate_domain([p[0] for p in pts], [p[1] for p in pts]).degree() == 1
    for trial in range(120):
        xs = [rnd() for _ in range(3)]
        a, b = rnd(), rnd()
        ys = [[a * x + b for x in xs], [b for _ in xs], [rnd() for _ in xs], [xf.zero() for _ in xs]][trial % 4]
        points = list(zip(xs, ys))
        assert sb.colinear(points) == slow(points) == (trial % 4 == 0)
