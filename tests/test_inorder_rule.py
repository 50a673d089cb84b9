"""The rule k_inorder implements (phanotate_amd/csrc/phx_inorder.inc), checked on the CPU against what it must reproduce:
the parents an in-place Bellman-Ford leaves when it relaxes the edges in list order with a strict '<' (the reference's
solver boundary as the golden generator states it, tests/golden/make_golden.py:100-133; call sites phanotate.py:56-64).

Claim: with final distances d, parent(v) is determined without replaying the relaxation.  T(source) = (0, -1);
a tight edge e = (u -> v) (d[u] + w == d[v]) fires at (round(u) + [e < parent_edge(u)], e); T(v) is the minimum over the
tight in-edges of v and parent(v) its arg min.  A node is final once all its tight in-neighbours are; when a zero-length
cycle of tight edges blocks that, the earliest firing edge out of the final set is right (Dijkstra step)."""
import random

from conftest import inorder_bellman_ford


def rule_parents(V, edges, s, dist):
    tin = [[] for _ in range(V)]
    for i, (u, v, w) in enumerate(edges):
        if v != s and dist[u] is not None and dist[v] is not None and dist[u] + w == dist[v]:
            tin[v].append(i)
    T = [None] * V
    par = [-1] * V
    T[s] = (0, -1)

    def fire(i):
        r, k = T[edges[i][0]]
        return (r + (1 if i < k else 0), i)

    pending = {v for v in range(V) if dist[v] is not None and v != s}
    while pending:
        ready = [v for v in pending if tin[v] and all(T[edges[i][0]] is not None for i in tin[v])]
        if ready:
            new = [(v, min(tin[v], key=fire)) for v in ready]
        else:
            cands = [(fire(i), i) for v in pending for i in tin[v] if T[edges[i][0]] is not None]
            assert cands
            i = min(cands)[1]
            new = [(edges[i][1], i)]
        for v, i in new:
            T[v] = fire(i)
            par[v] = i
            pending.discard(v)
    return par


def test_parent_rule_equals_in_place_bellman_ford():
    rng = random.Random(1)
    n = 0
    while n < 4000:
        V = rng.randint(3, 14)
        edges, seen = [], set()
        for _ in range(rng.randint(V, 4 * V)):
            u, v = rng.randrange(V), rng.randrange(V)
            if u == v or (u, v) in seen:
                continue
            seen.add((u, v))
            edges.append((u, v, rng.choice([0, 0, 1, 1, 2, 3, -1]) if rng.random() < 0.9 else rng.randint(-2, 5)))
        dist, par = inorder_bellman_ford(V, edges, 0)
        if dist is None:
            continue  # negative cycle
        n += 1
        got = rule_parents(V, edges, 0, dist)
        for v in range(1, V):
            if dist[v] is not None:
                assert got[v] == par[v], (V, edges, v)
