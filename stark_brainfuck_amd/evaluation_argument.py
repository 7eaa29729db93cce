"""Evaluation arguments: what the verifier recomputes from public data -- the input and output symbols, the program -- and
compares with the terminals the prover sent.  Host-side mirror of /root/reference/code/evaluation_argument.py:1-53;
values are int triples (air.x*).

Both arguments are Horner evaluations in a challenge: of the symbol string in iota (input: gamma, output: delta), and of the
compressed program rows a*address + b*instruction + c*next_instruction in eta (the instruction table's evaluation column)."""
from functools import reduce

from .air import X0, xadd, xlift, xmul, xscale


def _value(x):
    return x.value if hasattr(x, "value") else int(x)


def _horner(terms, point):
    return reduce(lambda acc, term: xadd(xmul(acc, point), term), terms, X0)


class _TerminalCheck:
    terminal_index = None

    def select_terminal(self, terminals):
        return terminals[self.terminal_index]


class EvaluationArgument(_TerminalCheck):
    def __init__(self, challenge_index, terminal_index, symbols):
        self.challenge_index, self.terminal_index, self.symbols = challenge_index, terminal_index, symbols

    def compute_terminal(self, challenges):
        return _horner([xlift(_value(s)) for s in self.symbols], challenges[self.challenge_index])


class ProgramEvaluationArgument(_TerminalCheck):
    def __init__(self, challenge_indices, terminal_index, program):
        self.challenge_indices, self.terminal_index, self.program = challenge_indices, terminal_index, program

    def compute_terminal(self, challenges):
        a, b, c, eta = (challenges[i] for i in sorted(self.challenge_indices))
        words = [_value(w) for w in self.program]
        following = words[1:] + [0]
        rows = [xadd(xadd(xscale(a, address), xscale(b, word)), xscale(c, nxt))
                for address, (word, nxt) in enumerate(zip(words, following))]
        # the reference appends one more row for the address just behind the program: instruction 0, next instruction 0
        rows.append(xscale(a, len(words)))
        return _horner(rows, eta)
