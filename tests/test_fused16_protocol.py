"""The ticket protocol of fused16_kernel (ntt_amd/csrc/kernels.hpp), modelled step by step in Python
and run under random interleavings: every transform gets each of its 16 first-phase and 16
second-phase tiles exactly once, all tiles of a transform are handled on one XCD, a second-phase tile
never starts before the transform's 16 first-phase tiles are stored, team indices stay inside the
array the host sizes for them, and every workgroup terminates -- for any number of workgroups per
XCD, XCDs without workgroups included."""
import random

import pytest


def workgroup(st, xcc, log):
    """One workgroup as a generator; every yield is a point where other workgroups may run."""
    batch = st["batch"]
    more_first, holding, T2, m2 = True, False, 0, 0
    while True:
        ready = False
        if holding:
            ready = st["arrived"][T2] >= 16
            yield
        if not ready and (more_first or not holding):
            t = st["ticket1"][xcc]
            st["ticket1"][xcc] += 1
            yield
            k, m = t >> 4, t & 15
            assert k < st["team_stride"], "team index outside the array"
            if m == 0:
                while k > 0 and st["team"][xcc][k - 1] == 0:
                    yield
                claimed = st["next"]
                st["next"] += 1
                v = min(claimed, batch) + 1
                st["team"][xcc][k] = v
            else:
                while st["team"][xcc][k] == 0:
                    yield
                v = st["team"][xcc][k]
            T1 = v - 1
            if T1 < batch:
                yield  # the tile's work
                log.append(("first", T1, m, xcc))
                st["arrived"][T1] += 1
            else:
                more_first = False
        if not holding:
            t = st["ticket2"][xcc]
            st["ticket2"][xcc] += 1
            yield
            k = t >> 4
            assert k < st["team_stride"]
            spins = 0
            while st["team"][xcc][k] == 0:
                spins += 1
                assert spins < 100000, "second-phase ticket names a team nobody creates"
                yield
            T2, m2 = st["team"][xcc][k] - 1, t & 15
            if T2 >= batch:
                return
            holding = True
            continue
        if ready:
            assert st["arrived"][T2] == 16
            yield
            log.append(("second", T2, m2, xcc))
            holding = False


@pytest.mark.parametrize("batch,per_xcd", [(1, [1, 0, 0, 0, 0, 0, 0, 0]), (5, [1, 1, 1, 1, 1, 1, 1, 1]),
                                           (7, [3, 0, 17, 2, 0, 40, 1, 5]), (64, [64] * 8), (33, [16] * 8),
                                           (20, [2, 2, 2, 2, 2, 2, 2, 2]), (9, [0, 0, 0, 0, 0, 0, 0, 31])])
def test_protocol_under_random_interleavings(batch, per_xcd):
    for seed in range(6):
        rng = random.Random(seed * 1000 + batch)
        grid = sum(per_xcd)
        stride = batch + grid // 4 + 16                      # fused16_words() in nttx.hip
        st = {"batch": batch, "ticket1": [0] * 8, "ticket2": [0] * 8, "next": 0, "arrived": [0] * batch,
              "team": [[0] * stride for _ in range(8)], "team_stride": stride}
        log = []
        live = [workgroup(st, x, log) for x in range(8) for _ in range(per_xcd[x])]
        steps = 0
        while live:
            steps += 1
            assert steps < 5_000_000, "no termination"
            g = rng.choice(live) if rng.random() < 0.9 else live[0]
            try:
                next(g)
            except StopIteration:
                live.remove(g)
        first = sorted((T, m) for kind, T, m, _ in log if kind == "first")
        second = sorted((T, m) for kind, T, m, _ in log if kind == "second")
        want = [(T, m) for T in range(batch) for m in range(16)]
        assert first == want and second == want
        home = {}
        for _, T, _, x in log:
            assert home.setdefault(T, x) == x, "a transform's tiles ran on two XCDs"
        done_first = {}
        for kind, T, m, _ in log:                              # log order = completion order
            if kind == "first":
                done_first[T] = done_first.get(T, 0) + 1
            else:
                assert done_first.get(T, 0) == 16
