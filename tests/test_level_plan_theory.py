"""The argument behind the level-synchronous grower, as an executable check (CPU, no library involved).

csrc/rgbm_level.h grows LightGBM's leaf-wise (best-first) tree from level-synchronous passes.  Before the pass of a level
`k_level_plan` decides which nodes of that level to split speculatively:

    pm(X)  = min over the path root..X of the nodes' best split gains
    expand X  iff  gain(X) > 0  and  fewer than num_leaves-1 KNOWN nodes Y have pm(Y) > pm(X)

(known = every node of this or a shallower level that exists in the speculative tree).  `k_level_replay` then runs
LightGBM's own best-first loop over the speculative nodes.  That is only right if best-first never wants to split a node
the rule left unexpanded.  This file restates both procedures over random "gain trees" (every potential node of a depth-7
tree gets a random best gain, ties and non-positive gains included) and checks that containment for thousands of trees
and budgets -- including the tie-break of LightGBM's ArgMax (first maximum = smallest leaf index).
"""
import numpy as np
import pytest


def best_first(gain, max_depth, num_leaves):
    """LightGBM SerialTreeLearner::Train over a complete binary gain tree (heap numbering, root = 1).
    Returns the list of split nodes in split order."""
    leaf_node = [1]                      # leaf index -> heap node id
    splits = []

    def g(node):
        depth = node.bit_length() - 1
        return gain[node] if depth < max_depth else -np.inf

    for _ in range(num_leaves - 1):
        gains = [g(n) for n in leaf_node]
        best = int(np.argmax(gains))     # first maximum, like ArrayArgs::ArgMax
        if not gains[best] > 0.0:
            break
        node = leaf_node[best]
        splits.append(node)
        leaf_node[best] = 2 * node       # the left child keeps the leaf index
        leaf_node.append(2 * node + 1)   # the right child gets the next one
    return splits


def level_synchronous(gain, max_depth, num_leaves):
    """k_level_plan's rule, level by level.  Returns the set of expanded (speculatively split) nodes."""
    pm = {}
    level = [1]
    expanded = set()
    for depth in range(max_depth):
        for n in level:
            gn = gain[n]
            v = gn if n == 1 else min(pm[n // 2], gn)
            pm[n] = v if gn > -np.inf else -np.inf
        known = np.array(list(pm.values()))
        nxt = []
        for n in level:
            v = pm[n]
            if v > 0.0 and int((known > v).sum()) < num_leaves - 1:
                expanded.add(n)
                nxt += [2 * n, 2 * n + 1]
        level = nxt
        if not level:
            break
    return expanded


@pytest.mark.parametrize("kind", ["continuous", "ties", "sparse", "decaying"])
def test_best_first_never_leaves_the_speculative_tree(kind):
    rng = np.random.default_rng({"continuous": 1, "ties": 2, "sparse": 3, "decaying": 4}[kind])
    n_nodes = 1 << 8                     # depths 0..7
    checked = 0
    for trial in range(4000):
        max_depth = int(rng.integers(1, 8))
        num_leaves = int(rng.integers(2, 129))
        if kind == "continuous":
            gain = rng.gamma(1.0, 1.0, n_nodes)
        elif kind == "ties":
            gain = rng.integers(0, 4, n_nodes).astype(np.float64)          # many equal gains, zeros included
        elif kind == "sparse":
            gain = np.where(rng.random(n_nodes) < 0.35, -np.inf, rng.gamma(1.0, 1.0, n_nodes))
        else:                                                               # gains shrink with depth, as real trees do
            depth = np.floor(np.log2(np.maximum(np.arange(n_nodes), 1)))
            gain = rng.gamma(1.0, 1.0, n_nodes) * 0.6 ** depth
            gain[rng.random(n_nodes) < 0.1] = 0.0
        splits = best_first(gain, max_depth, num_leaves)
        expanded = level_synchronous(gain, max_depth, num_leaves)
        missing = [n for n in splits if n not in expanded]
        assert not missing, "trial %d (%s): best-first splits %r outside the speculative tree (max_depth=%d, num_leaves=%d)" % (
            trial, kind, missing, max_depth, num_leaves)
        # the speculation stays bounded: at most num_leaves-1 .. 2x that many expansions per level by construction,
        # and never a node below an unexpanded parent
        assert all(n == 1 or (n // 2) in expanded for n in expanded)
        checked += len(splits)
    assert checked > 1000


def test_the_budget_bound_is_tight_enough_to_prune():
    """With a small leaf budget much of a deep gain tree must stay unexpanded (otherwise the passes would build
    histograms for nothing).  31 leaves on a full depth-7 tree (127 inner nodes, 30 of them split by best-first): with
    gains that shrink with depth, as real trees have, well under half of the inner nodes are expanded; even with i.i.d.
    gains (the worst case for a path-minimum bound) a good third is pruned."""
    rng = np.random.default_rng(9)
    depth = np.floor(np.log2(np.maximum(np.arange(256), 1)))
    frac_iid, frac_decay = [], []
    for _ in range(300):
        gain = rng.gamma(1.0, 1.0, 256)
        frac_iid.append(len(level_synchronous(gain, 7, 31)) / 127.0)
        assert len(best_first(gain, 7, 31)) == 30
        frac_decay.append(len(level_synchronous(gain * 0.6 ** depth, 7, 31)) / 127.0)
    assert np.mean(frac_iid) < 0.65
    assert np.mean(frac_decay) < 0.45
