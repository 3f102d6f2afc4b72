"""CPU restatement of the reference's prioritised replay (BaseClass/replay_buffer.py:57-223).  TEST INFRASTRUCTURE:
only tests/ may import this; the product path is csrc/per.hip.  Pinned against tests/golden/per.npz, which
oracle/gen_golden_per.py makes by executing the reference (tests/test_oracle_golden.py::test_per_*).

What is restated, with the reference's quirks:
  * the sum tree is a flat array of 2c-1 nodes, leaf of data slot d at node c-1+d (:60-66, :84-90); priorities
    reach the inner nodes as propagated DIFFERENCES (:70-79), so an inner node is a history-dependent float sum;
  * the descent goes left when v <= tree[left] (:103-110); with a capacity that is not a power of two the leaves
    are therefore visited in a rotated slot order;
  * total() truncates the root to an int (:117-118) and that int is what segments and probabilities use (:147,:163);
  * priority on push: (|error| + 0.01) ** 0.6, no clip (:143-144); on batch_update: min(|error| + 0.01, 1) ** 0.6
    (:215-222); beta <- min(1, beta + 0.001) on every sample() (:155); weights (n_entries * p / total) ** -beta,
    divided by their maximum (:176-178).
"""
import numpy as np


class PerOracle:
    def __init__(self, capacity, alpha=0.6, beta=0.4, beta_inc=0.001, epsilon=0.01, clip=1.0):
        self.c = int(capacity)
        self.node = np.zeros(2 * self.c - 1)          # :66
        self.ptr = 0
        self.n_entries = 0
        self.alpha, self.beta, self.beta_inc, self.epsilon, self.clip = alpha, beta, beta_inc, epsilon, clip

    # ---- SumTree.update / add (:68-97)
    def _set_leaf(self, node_idx, p):
        delta = p - self.node[node_idx]
        self.node[node_idx] = p
        k = node_idx
        while k != 0:
            k = (k - 1) // 2
            self.node[k] += delta

    def push(self, error):
        p = (abs(float(error)) + self.epsilon) ** self.alpha                 # :143
        slot = self.ptr
        self._set_leaf(slot + self.c - 1, p)
        self.ptr = (self.ptr + 1) % self.c
        self.n_entries = min(self.n_entries + 1, self.c)
        return slot

    def batch_update(self, slots, abs_errors):
        e = np.minimum(np.asarray(abs_errors, dtype=np.float64) + self.epsilon, self.clip)   # :216-218
        for s, p in zip(slots, np.power(e, self.alpha)):
            self._set_leaf(int(s) + self.c - 1, p)

    def priorities(self):
        return self.node[self.c - 1:].copy()

    def total_int(self):
        return int(self.node[0])                                             # :117-118

    # ---- SumTree.get_leaf (:99-115)
    def _descend(self, v):
        k = 0
        while 2 * k + 1 < len(self.node):
            left = 2 * k + 1
            if v <= self.node[left]:
                k = left
            else:
                v -= self.node[left]
                k = left + 1
        return k

    def segments(self, batch):
        seg = self.total_int() / batch                                       # :147
        return seg * np.arange(batch), seg * (np.arange(batch) + 1)

    def sample(self, batch, draws):
        """draws[i] is what random.uniform(seg*i, seg*(i+1)) returned.  -> (slots, priorities, weights)"""
        self.beta = min(1.0, self.beta + self.beta_inc)                      # :155
        leaves = np.array([self._descend(float(v)) for v in draws])
        p = self.node[leaves]
        prob = p / self.total_int()                                          # :175
        w = np.power(self.n_entries * prob, -self.beta)                      # :176
        w /= w.max()
        return leaves - (self.c - 1), p, w


def leaf_rotation(capacity):
    """In-order position q of the flat tree's leaves holds data slot (q + rot) % capacity."""
    depth = int(np.floor(np.log2(2 * capacity - 1)))
    return (1 << depth) - capacity


def sample_by_cumsum(prio, total_for_nothing, draws):
    """The same selection without a tree: first in-order position whose inclusive cumulative priority reaches v
    (what csrc/per.hip does).  Agrees with the descent except when v falls within rounding of a boundary."""
    c = len(prio)
    rot = leaf_rotation(c)
    order = (np.arange(c) + rot) % c
    cs = np.cumsum(prio[order])
    q = np.searchsorted(cs, np.asarray(draws), side="left")
    return order[np.minimum(q, c - 1)]
