"""Protocol model of tools/probe/g256p_probe.hip's LDS-DMA ring (no GPU needed): the order of DMA batches, fragment reads, counted
waits and barriers of one workgroup walking several tiles, exactly as the kernel issues them, checked for the two things a ring can get
wrong whatever the timing:

  (a) a slot is refilled only after every wave has finished reading its previous content -- i.e. at least one barrier lies between the
      last read of (tile, K-step) in a slot and the issue of the DMA batch that overwrites it (reads are complete when a wave arrives at
      a barrier: end_step waits lgkmcnt(0));
  (b) a slot is read only after the batch that fills it has been retired by a counted wait of EVERY wave -- i.e. a `vmcnt(8)` (all but
      the eight youngest DMA instructions = all but the batch issued in the same K-step) and a barrier lie between the issue and the read.

The kernel's order per K-step g of a tile (substep): [2 trickle stores] issue(g+3) interleaved with mma(g) and the reads of g+1, then
vmcnt(8) lgkmcnt(0), barrier.  K-step 0 of a tile (first_step): conversion, reads of K-step 0, lgkmcnt(0) + barrier, issue(g+3) with
mma(0) and the reads of K-step 1, end_step.  The last K-step of a tile reads nothing ahead."""


def run(tiles, nk):
    nsub = 2 * nk
    NSLOT = 3
    events = []                       # ("issue", key, slot) | ("read", key, slot) | ("wait",) | ("barrier",)
    issue_q = [(t, ks) for t in range(tiles) for ks in range(nsub)] + [("dead", i) for i in range(3)]
    state = {"ii": 0, "islot": 0, "rslot": 0, "ri": 0}
    read_q = [(t, ks) for t in range(tiles) for ks in range(nsub)]

    def issue():
        events.append(("issue", issue_q[state["ii"]], state["islot"]))
        state["ii"] += 1
        state["islot"] = (state["islot"] + 1) % NSLOT

    def read():
        events.append(("read", read_q[state["ri"]], state["rslot"]))
        state["ri"] += 1
        state["rslot"] = (state["rslot"] + 1) % NSLOT

    def end_step():
        events.append(("wait",))
        events.append(("barrier",))

    # prologue
    issue(); issue(); issue()
    events.append(("wait0",))         # vmcnt(0)
    events.append(("barrier",))
    for t in range(tiles):
        # first_step(true)
        read()                        # K-step 0 of this tile
        events.append(("barrier",))   # lgkmcnt(0) + barrier
        issue()
        read()                        # K-step 1 rolls in
        end_step()
        for ks in range(1, nsub - 1):
            issue()
            read()                    # K-step ks + 1
            end_step()
        issue()                       # last K-step: no read ahead
        end_step()
    # checks
    content = {}                      # slot -> key currently (being) written
    issued_at = {}                    # key -> event index of its issue
    last_read_of_slot = {}            # slot -> event index of the last read
    barrier_idx = [i for i, e in enumerate(events) if e[0] == "barrier"]
    # retired_at[key] = index of the barrier that follows the first counted wait retiring it
    n_issued = 0
    pending = []                      # keys in issue order, not yet retired
    retired_at = {}
    for i, e in enumerate(events):
        if e[0] == "issue":
            key, slot = e[1], e[2]
            # (a) at least one barrier between the last read of this slot and this issue
            if slot in last_read_of_slot:
                assert any(last_read_of_slot[slot] < b < i for b in barrier_idx), ("refill without a barrier", key, slot)
            issued_at[key] = i
            content[slot] = key
            pending.append(key)
        elif e[0] == "wait":          # vmcnt(8): everything but the youngest batch
            for k in pending[:-1]:
                retired_at.setdefault(k, next(b for b in barrier_idx if b > i))
            pending = pending[-1:]
        elif e[0] == "wait0":
            for k in pending:
                retired_at.setdefault(k, next(b for b in barrier_idx if b > i))
            pending = []
        elif e[0] == "read":
            key, slot = e[1], e[2]
            assert content.get(slot) == key, ("slot holds something else", key, slot, content.get(slot))
            assert key in retired_at and retired_at[key] < i, ("read before the batch was retired and published", key)
            last_read_of_slot[slot] = i
    assert state["ri"] == tiles * nsub
    return len(events)


if __name__ == "__main__":
    for tiles in (1, 2, 3, 5):
        for nk in (10, 12, 13, 36):
            run(tiles, nk)
    print("g256p ring protocol OK")
