"""Input / output tables -- mirror of the reference's `io_table.py` (/root/reference/code/io_table.py): padding that
redefines length and height (:17-21) and the running evaluation (:77-110).  Constraints: air.IOAir."""
import numpy as np

from . import air
from .air import xpow, X0
from .table import Table


class IOTable(Table):
    column, evaluation = 0, 1

    def __init__(self, field, length, generator, order):
        super().__init__(field, 1, 2, length, 0, generator, order)

    def pad(self):
        self.length, _ = self._rows_and_last()
        self._pad_to(np.zeros((1, self._padding_length(self.length)), dtype=np.uint64))
        self.height = len(self.matrix)

    def air_params(self, challenges):
        return [xpow(tuple(challenges[self.challenge_index]), self.height - self.length)]

    def _scans(self, all_challenges, all_initials):
        """io_table.py:77-110: evaluation = evaluation * iota + symbol on every row"""
        return [dict(kind=1, cols=[0], mask=None, constants=[all_challenges[self.challenge_index], (1, 0, 0)], initial=X0, before=False)]

    def _terminal_reads(self):
        return [(0, self.length - 1)] if self.length else []

    def _after_extend(self, terminals, all_challenges, read):
        # the terminal is the value after the last real (unpadded) row
        self.evaluation_terminal = read(0, self.length - 1) if self.length else X0


class InputTable(IOTable):
    air = air.TABLE_AIRS[3]
    table_index = 3
    challenge_index, terminal_index = 8, 2


class OutputTable(IOTable):
    air = air.TABLE_AIRS[4]
    table_index = 4
    challenge_index, terminal_index = 9, 3
