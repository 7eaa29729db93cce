"""Processor table -- mirror of the reference's `processor_table.py` (/root/reference/code/processor_table.py):
column names, padding (:24-35) and the running products / evaluations of `extend` (:329-427).  The constraints
themselves live in air.ProcessorAir."""
from . import air
from .air import xadd, xmul, xsub, xlift, X0
from .table import Table, P


class ProcessorTable(Table):
    cycle, instruction_pointer, current_instruction, next_instruction, memory_pointer, memory_value, memory_value_inverse = range(7)
    instruction_permutation, memory_permutation, input_evaluation, output_evaluation = 7, 8, 9, 10
    air = air.TABLE_AIRS[0]
    table_index = 0

    def __init__(self, field, length, num_randomizers, generator, order):
        super().__init__(field, 7, 11, length, num_randomizers, generator, order)

    def pad(self):
        rows = [list(r) for r in self.base_rows()]
        while len(rows) & (len(rows) - 1):
            last = rows[-1]
            rows.append([(last[0] + 1) % P, last[1], 0, 0, last[4], last[5], last[6]])
        self._append_rows(rows)

    def extend(self, all_challenges, all_initials):
        a, b, c, d, e, f, alpha, beta, gamma, delta, eta = all_challenges
        ipp, mpp = all_initials
        iev = oev = X0
        iev_id = oev_id = None
        rows, ext = self.base_rows(), []
        for i, row in enumerate(rows):
            clk, ip, ci, ni, mp, mv, _ = row
            ext.append([ipp, mpp, iev, oev])
            if ci != 0:
                ipp = xmul(ipp, xsub(xsub(xsub(alpha, air.xscale(a, ip)), air.xscale(b, ci)), air.xscale(c, ni)))
                mpp = xmul(mpp, xsub(xsub(xsub(beta, air.xscale(d, clk)), air.xscale(e, mp)), air.xscale(f, mv)))
            if ci == ord(","):      # the input symbol shows up in the NEXT row's memory value
                iev, iev_id = _evaluation_step(iev, iev_id, gamma, self.matrix[i + 1][5])
            if ci == ord("."):
                oev, oev_id = _evaluation_step(oev, oev_id, delta, self.matrix[i][5])
        self.ext_rows = ext
        self.instruction_permutation_terminal = ipp
        self.memory_permutation_terminal = mpp
        self.input_evaluation_terminal = iev
        self.output_evaluation_terminal = oev
        self.evaluation_terminal_identities = (iev_id, oev_id)


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
