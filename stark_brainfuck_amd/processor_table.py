"""Processor table -- mirror of the reference's `processor_table.py` (/root/reference/code/processor_table.py):
column names, padding (:24-35) and the running products / evaluations of `extend` (:329-427).  The constraints
themselves live in air.ProcessorAir."""
import numpy as np

from . import air
from .air import xadd, xmul, xlift, X0
from .table import Table, P


class ProcessorTable(Table):
    cycle, instruction_pointer, current_instruction, next_instruction, memory_pointer, memory_value, memory_value_inverse = range(7)
    instruction_permutation, memory_permutation, input_evaluation, output_evaluation = 7, 8, 9, 10
    air = air.TABLE_AIRS[0]
    table_index = 0

    def __init__(self, field, length, num_randomizers, generator, order):
        super().__init__(field, 7, 11, length, num_randomizers, generator, order)

    def pad(self):
        rows, last = self._rows_and_last()
        k = self._padding_length(rows)
        pad = np.zeros((7, k), dtype=np.uint64)
        pad[0] = self._counting(last[0], k)                            # the cycle count keeps counting (processor_table.py:24-35)
        for col in (1, 4, 5, 6):                                       # instruction pointer, memory pointer / value / inverse stay
            pad[col] = last[col]
        self._pad_to(pad)

    def _make_scan_masks(self):
        ci = self.base_array()[2]
        active = ci != 0                                                   # padding rows leave the products alone
        return [active, active, ci == ord(","), ci == ord(".")]

    def _scans(self, all_challenges, all_initials):
        """the four extension columns as running products / evaluations (processor_table.py:329-427)"""
        a, b, c, d, e, f, alpha, beta, gamma, delta, eta = all_challenges
        active, _, self._reads, self._writes = self._scan_masks()
        one = (1, 0, 0)
        return [dict(kind=0, cols=[1, 2, 3], mask=active, constants=[alpha, a, b, c], initial=all_initials[0], before=True),
                dict(kind=0, cols=[0, 4, 5], mask=active, constants=[beta, d, e, f], initial=all_initials[1], before=True),
                # an input symbol shows up in the NEXT row's memory value
                dict(kind=1, cols=[5], shift1=1, mask=self._reads, constants=[gamma, one], initial=X0, before=True),
                dict(kind=1, cols=[5], mask=self._writes, constants=[delta, one], initial=X0, before=True)]

    def _after_extend(self, terminals, all_challenges, read):
        gamma, delta = all_challenges[8], all_challenges[9]
        t_ipp, t_mpp, t_iev, t_oev = terminals
        self.instruction_permutation_terminal = t_ipp
        self.memory_permutation_terminal = t_mpp
        self.input_evaluation_terminal = t_iev
        self.output_evaluation_terminal = t_oev
        self.evaluation_terminal_identities = (self._identity(None, t_iev, gamma, np.nonzero(self._reads)[0] + 1),
                                               self._identity(None, t_oev, delta, np.nonzero(self._writes)[0]))

    def _identity(self, states, terminal, challenge, symbol_rows):
        """what the reference's OBJECT of this terminal is made of, replaying _evaluation_step over the update rows"""
        state, identity = X0, None
        for r in symbol_rows:
            state, identity = _evaluation_step(state, identity, challenge, self.matrix[int(r)][5])
        assert tuple(state) == tuple(terminal)
        return identity


def _evaluation_step(state, identity, challenge, symbol):
    """state * challenge + lift(symbol) on int triples, tracking what the reference's OBJECTS look like
    (processor_table.py:390-404 with univariate.py:23-35): while the running value is zero the sum IS the lifted symbol's
    polynomial -- the very BaseFieldElement object from the matrix; afterwards every result is made of new elements that
    point at the BaseField instance of the left operand's coefficients.  identity: None (no coefficients), ("object", element)
    or ("fresh", base_field_instance); pickle shows the difference (memoisation by identity, one copy per field instance)."""
    value = symbol.value if hasattr(symbol, "value") else int(symbol)
    product = xmul(state, challenge)
    result = xadd(product, xlift(value))
    if not any(result):
        return result, None
    if not any(product):
        return result, (("object", symbol) if hasattr(symbol, "field") else None)
    if identity is None:
        return result, None
    return result, ("fresh", identity[1].field if identity[0] == "object" else identity[1])
