"""Input / output tables -- mirror of the reference's `io_table.py` (/root/reference/code/io_table.py): padding that
redefines length and height (:17-21) and the running evaluation (:77-110).  Constraints: air.IOAir."""
from . import air
from .air import xadd, xmul, xlift, xpow, X0
from .table import Table


class IOTable(Table):
    column, evaluation = 0, 1

    def __init__(self, field, length, generator, order):
        super().__init__(field, 1, 2, length, 0, generator, order)

    def pad(self):
        rows = [list(r) for r in self.base_rows()]
        self.length = len(rows)
        while len(rows) & (len(rows) - 1):
            rows.append([0])
        self._append_rows(rows)
        self.height = len(rows)

    def air_params(self, challenges):
        return [xpow(tuple(challenges[self.challenge_index]), self.height - self.length)]

    def extend(self, all_challenges, all_initials):
        iota = all_challenges[self.challenge_index]
        running = terminal = X0
        ext = []
        for i, (v,) in enumerate(self.base_rows()):
            running = xadd(xmul(running, iota), xlift(v))
            ext.append([running])
            if i == self.length - 1:
                terminal = running
        self.ext_rows = ext
        self.evaluation_terminal = terminal


class InputTable(IOTable):
    air = air.TABLE_AIRS[3]
    table_index = 3
    challenge_index, terminal_index = 8, 2


class OutputTable(IOTable):
    air = air.TABLE_AIRS[4]
    table_index = 4
    challenge_index, terminal_index = 9, 3
