"""CPU model of what a launch of several traversals relies on (csrc/bfs_persist.hip: bfs_persistent_kernel<T>, T < 1024),
restated in Python and run under random interleavings of the workgroups:

* the launch's counter: sub-grid j starts on traversal j, a sub-grid that finishes one draws n_grids + (old - base) from
  the monotonic counter -- every traversal of the table is run exactly once, every sub-grid's last draw fails, and the
  counter ends at base + ntrav (what the host adds to its copy after a chained launch);
* the three state blocks: traversal number t of a sub-grid runs on block t % 3 and clears block (t + 1) % 3 at its start;
  a workgroup passes a traversal's end barrier only when every workgroup of the sub-grid has arrived there.  Whenever a
  workgroup clears a block, no workgroup of the sub-grid is still inside the traversal that last ran on it (t - 2) -- its
  barrier counters live in that block -- and the block a traversal runs on has been cleared completely before anybody
  is in it.
No GPU; the device code is tested against the oracle in tests/test_gpu_algorithms.py::test_bfs_coscheduled."""
import random


def run_launch(rng, n_grids, ntrav, wgs, base):
    """one launch: returns (order in which traversals were run per sub-grid, final counter)"""
    counter = base
    chained = ntrav > n_grids
    # per sub-grid state
    rot = [rng.randrange(0, 5) for _ in range(n_grids)]          # traversals the sub-grid ran before this launch
    ran = [[] for _ in range(n_grids)]
    # a workgroup: (traversal index in the table or None, stage) ; stages: 0 clear, 1 work, 2 arrived at the end barrier
    cur = [j if j < ntrav else None for j in range(n_grids)]      # the traversal the sub-grid is on (same for its workgroups)
    pos = [[(0, 0)] * wgs for _ in range(n_grids)]               # (local traversal number since launch start, stage)
    nxt = [None] * n_grids                                       # published by workgroup 0 before it arrives
    arrived = [0] * n_grids                                      # monotonic, as the device barrier's counters are
    block_user = [dict() for _ in range(n_grids)]                # block -> workgroups currently inside a traversal on it
    prepared = [dict() for _ in range(n_grids)]                  # block -> [traversal number it was cleared for, workgroups that cleared]
    table = [[j] for j in range(n_grids)]                        # traversal run at local number k
    done = [cur[j] is None for j in range(n_grids)]
    for j in range(n_grids):
        for b in range(3):
            block_user[j][b] = set()
        for t0 in range(rot[j], rot[j] + 3):                     # the host cleared everything before the first launch
            prepared[j][t0 % 3] = [t0, set(range(wgs))]
    steps = 0
    while not all(done):
        steps += 1
        assert steps < 10 ** 6
        j = rng.choice([x for x in range(n_grids) if not done[x]])
        w = rng.randrange(wgs)
        k, stage = pos[j][w]
        if k >= len(table[j]):
            continue                                             # this workgroup has left the launch
        t = rot[j] + k                                           # the sub-grid's traversal number
        if stage == 0:
            # about to enter traversal t: its block must be completely cleared, and nobody may still be inside the
            # traversal that last ran on the block it is going to clear
            assert prepared[j][t % 3] == [t, set(range(wgs))], "a traversal starts on a block that is not clean"
            assert not block_user[j][(t + 1) % 3], "a block is cleared while a workgroup is still inside its traversal"
            block_user[j][t % 3].add(w)
            if prepared[j][(t + 1) % 3][0] != t + 1:
                prepared[j][(t + 1) % 3] = [t + 1, set()]
            prepared[j][(t + 1) % 3][1].add(w)
            pos[j][w] = (k, 1)
        elif stage == 1:
            if not chained:
                block_user[j][t % 3].discard(w)
                pos[j][w] = (k + 1, 0)
                if all(p[0] > k for p in pos[j]):
                    ran[j].append(table[j][k])
                    done[j] = True
                continue
            if w == 0:
                nxt[j] = (k, n_grids + (counter - base))          # (the block's word: read by everybody after THIS barrier)
                counter += 1
            arrived[j] += 1
            pos[j][w] = (k, 2)
        else:
            if arrived[j] < wgs * (k + 1):
                continue                                         # the barrier (monotonic, generation k + 1) has not completed
            block_user[j][t % 3].discard(w)
            if len(table[j]) == k + 1 and not (ran[j] and len(ran[j]) > k):   # first workgroup through the barrier fixes the next entry
                assert nxt[j] is not None and nxt[j][0] == k
                ran[j].append(table[j][k])
                if nxt[j][1] < ntrav:
                    table[j].append(nxt[j][1])
            pos[j][w] = (k + 1, 0)
            if all(p[0] > k for p in pos[j]) and len(table[j]) == k + 1:
                done[j] = True
    return ran, counter


def test_counter_and_block_rotation_under_random_interleavings():
    rng = random.Random(7)
    for _ in range(300):
        n_grids = rng.randrange(1, 9)
        ntrav = rng.randrange(n_grids, 30)
        wgs = rng.randrange(1, 6)
        base = rng.randrange(0, 1000)
        ran, counter = run_launch(rng, n_grids, ntrav, wgs, base)
        flat = sorted(x for r in ran for x in r)
        assert flat == list(range(ntrav)), (n_grids, ntrav, ran)             # every traversal exactly once
        if ntrav > n_grids:
            assert counter == base + ntrav                                    # one draw per traversal: what the host adds
        else:
            assert counter == base
