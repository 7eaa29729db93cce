"""Evaluation arguments -- mirror of the reference's `evaluation_argument.py`
(/root/reference/code/evaluation_argument.py:1-53): what the verifier recomputes from public data (input and output
symbols, the program) and compares with the terminals.  Values are int triples (air.x*)."""
from .air import X0, xadd, xmul, xlift, xscale


def _v(x):
    return x.value if hasattr(x, "value") else int(x)


class EvaluationArgument:
    def __init__(self, challenge_index, terminal_index, symbols):
        self.challenge_index = challenge_index
        self.terminal_index = terminal_index
        self.symbols = symbols

    def compute_terminal(self, challenges):
        iota = challenges[self.challenge_index]
        acc = X0
        for s in self.symbols:
            acc = xadd(xmul(iota, acc), xlift(_v(s)))
        return acc

    def select_terminal(self, terminals):
        return terminals[self.terminal_index]


class ProgramEvaluationArgument:
    def __init__(self, challenge_indices, terminal_index, program):
        self.challenge_indices = challenge_indices
        self.terminal_index = terminal_index
        self.program = program

    def compute_terminal(self, challenges):
        a, b, c, eta = [challenges[i] for i in range(len(challenges)) if i in self.challenge_indices]
        words = [_v(p) for p in self.program] + [0]
        running = X0
        for i in range(len(words) - 1):          # every address occurs once: the "address changed" test is always true
            running = xadd(xadd(xadd(xmul(running, eta), xscale(a, i)), xscale(b, words[i])), xscale(c, words[i + 1]))
        index = len(words) - 1
        running = xadd(xadd(xmul(running, eta), xscale(a, index)), xscale(b, words[index]))
        return running

    def select_terminal(self, terminals):
        return terminals[self.terminal_index]
