"""N>1 path on CPU (gloo, world_size 2): shard the reads of one sample, count per shard, all-reduce(sum) the
occurrence counters through bronko_amd.dist, THEN threshold + map.  Must equal the single-process oracle.
The same test shows why pileups themselves must not be reduced."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import helpers


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out_dir):
    from bronko_amd.dist import allreduce_counters, shard_bounds
    from oracle import oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ix = orc.Index.load(os.path.join(helpers.GOLDEN, "hpv.bkdb"))
        reads = helpers.hpv_reads(6000, seed=31, err=0.01)
        # common key space = every distinct k-mer of the sample (stand-in for the engine's counter plane layout)
        keys, _, _ = orc.count_kmers(21, reads, ci=1, cs=2 ** 62, cx=2 ** 62)
        keys = np.sort(keys)
        lo, hi = shard_bounds(len(reads), rank, world)
        km, ct, _ = orc.count_kmers(21, reads[lo:hi], ci=1, cs=2 ** 62, cx=2 ** 62)
        plane = np.zeros(len(keys), np.int64)
        plane[np.searchsorted(keys, km)] = ct.astype(np.int64)
        t = torch.from_numpy(plane)
        allreduce_counters(t)                                    # the one exchange step
        total = t.numpy().astype(np.uint64)
        keep = (total >= 3) & (total <= 1000000000)              # -ci3 / -cx on the TRUE (reduced) count
        pile = orc.Pileup(ix)
        orc.map_kmers(ix, keys[keep], np.minimum(total[keep], 1000000), pile)
        # the wrong way round: per-shard thresholds + map, pileups summed afterwards
        shard_pile = orc.sample_pileup(ix, [reads[lo:hi]])
        wrong = torch.from_numpy(shard_pile.fwd_depth.astype(np.int64))
        dist.all_reduce(wrong)
        if rank == 0:
            np.savez(os.path.join(out_dir, "r0.npz"), fd=pile.fwd_depth, rd=pile.rev_depth, fk=pile.fwd_nk, rk=pile.rev_nk,
                     stats=pile.stats, wrong=wrong.numpy().astype(np.uint64))
    finally:
        dist.destroy_process_group()


def test_shard_bounds():
    from bronko_amd.dist import shard_bounds
    for n in (0, 1, 7, 1000003):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, r, w) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            assert max(h - l for l, h in b) - min(h - l for l, h in b) <= 1


def test_counter_allreduce_then_finalize_equals_single_process(oracle, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "r0.npz"))
    ix = oracle.Index.load(os.path.join(helpers.GOLDEN, "hpv.bkdb"))
    want = oracle.sample_pileup(ix, [helpers.hpv_reads(6000, seed=31, err=0.01)])
    assert np.array_equal(got["fd"], want.fwd_depth) and np.array_equal(got["rd"], want.rev_depth)
    assert np.array_equal(got["fk"], want.fwd_nk) and np.array_equal(got["rk"], want.rev_nk)
    assert np.array_equal(got["stats"], want.stats)
    # summing per-shard pileups is NOT the pileup of the sample
    assert (got["wrong"] != want.fwd_depth).sum() > 100
    ix.close()


def _worker_shards(rank, world, port, out_dir):
    from bronko_amd.dist import combine_shard_results, reduce_scatter_typed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(100 + rank)
        plane = torch.randint(-5, 50, (64 * 19,), generator=g, dtype=torch.int64)
        part = plane.numel() // world
        grp = dist.new_group(ranks=list(range(world)))   # (a group of its own, as every engine in flight gets one)
        mine = plane.clone()
        recv = torch.empty(part, dtype=torch.int64)
        reduce_scatter_typed(plane, recv, rank, world, group=grp)
        mine[rank * part:(rank + 1) * part] = recv
        mine32 = plane.clone()
        recv32 = torch.empty(part, dtype=torch.int32)
        reduce_scatter_typed(plane.to(torch.int32), recv32, rank, world)   # 32-bit elements on the wire: same numbers (negative differences included)
        mine32[rank * part:(rank + 1) * part] = recv32.to(torch.int64)
        depth = torch.randint(0, 1000, (40,), generator=g, dtype=torch.int64)
        nk = torch.randint(0, 1000, (40,), generator=g, dtype=torch.int64)
        sums = torch.randint(0, 1000, (14,), generator=g, dtype=torch.int64)
        d2, n2, s2 = depth.clone(), nk.clone(), sums.clone()
        combine_shard_results(d2, n2, s2)
        torch.save({"plane": plane, "mine": mine, "mine32": mine32, "depth": depth, "nk": nk, "sums": sums, "d2": d2, "n2": n2, "s2": s2},
                   os.path.join(out_dir, "s%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_reduce_scatter_and_combine_helpers(tmp_path):
    """bronko_amd.dist.reduce_scatter_typed leaves each rank its own part summed (on the default group and on one of its own);
    combine_shard_results is max / sum / sum."""
    world = 2
    port = _free_port()
    mp.spawn(_worker_shards, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(str(tmp_path), "s%d.pt" % r)) for r in range(world)]
    total = got[0]["plane"] + got[1]["plane"]
    part = total.numel() // world
    for r in range(world):
        assert torch.equal(got[r]["mine"][r * part:(r + 1) * part], total[r * part:(r + 1) * part])
        assert torch.equal(got[r]["mine32"][r * part:(r + 1) * part], total[r * part:(r + 1) * part])
        assert torch.equal(got[r]["d2"], torch.maximum(got[0]["depth"], got[1]["depth"]))
        assert torch.equal(got[r]["n2"], got[0]["nk"] + got[1]["nk"])
        assert torch.equal(got[r]["s2"], got[0]["sums"] + got[1]["sums"])


def _worker_entries(rank, world, port, out_dir):
    from bronko_amd.dist import exchange_entries
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        g = torch.Generator().manual_seed(500 + rank)
        splits = [3 + rank, 0 if rank == 1 else 5, 2]                  # ragged groups, one of them empty
        keys = torch.randint(0, 2 ** 62, (sum(splits),), generator=g, dtype=torch.int64)
        cnts = torch.randint(1, 1000, (sum(splits),), generator=g, dtype=torch.int32)
        rk, rc = exchange_entries(keys, cnts, splits)
        torch.save({"keys": keys, "cnts": cnts, "splits": splits, "rk": rk, "rc": rc}, os.path.join(out_dir, "e%d.pt" % rank))
    finally:
        dist.destroy_process_group()


def test_kmer_table_entries_all_to_all(tmp_path):
    """bronko_amd.dist.exchange_entries (the exchange step of full_kmer_stats under a sharded finalize): group r of every rank
    arrives at rank r, in rank order, ragged and empty groups included (world_size 3, gloo)."""
    world = 3
    port = _free_port()
    mp.spawn(_worker_entries, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = [torch.load(os.path.join(str(tmp_path), "e%d.pt" % r)) for r in range(world)]
    for r in range(world):
        want_k, want_c = [], []
        for q in range(world):
            lo = sum(got[q]["splits"][:r])
            want_k.append(got[q]["keys"][lo:lo + got[q]["splits"][r]])
            want_c.append(got[q]["cnts"][lo:lo + got[q]["splits"][r]])
        assert torch.equal(got[r]["rk"], torch.cat(want_k)) and torch.equal(got[r]["rc"], torch.cat(want_c))


def _lanes16_pack(plane, v_off, world):
    """numpy restatement of the engine's 16-bit transport (bk_kernels.hip, xport_pack_kernel<16>) for ONE part holding the whole
    plane: E counts (elements below v_off) as four 8-bit digits, V elements biased by L = 32767 // world, two 16-bit lanes per
    int32 word."""
    lim = 32767 // world
    e = plane[:v_off].astype(np.uint64)
    lanes = np.zeros(4 * v_off + (len(plane) - v_off), np.uint16)
    for q in range(4):
        lanes[q:4 * v_off:4] = ((e >> np.uint64(8 * q)) & np.uint64(255)).astype(np.uint16)
    lanes[4 * v_off:] = (plane[v_off:] + lim).astype(np.uint16)
    if len(lanes) % 2:
        lanes = np.concatenate([lanes, np.zeros(1, np.uint16)])
    return lanes.view(np.int32).copy()


def _lanes16_unpack(words, n, v_off, world):
    lanes = words.view(np.uint16).astype(np.int64)
    e = sum(lanes[q:4 * v_off:4] << (8 * q) for q in range(4))
    v = lanes[4 * v_off:4 * v_off + (n - v_off)] - (32767 // world) * world
    return np.concatenate([e, v])


def _worker_lanes(rank, world, port, out_dir):
    from bronko_amd.dist import pick_width, reduce_scatter_typed
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(700 + rank)
        v_off, n = 40, 400
        lim = 32767 // world
        plane = np.concatenate([rng.integers(0, 2 ** 32, v_off, dtype=np.int64),              # E counts: anything below 2^32
                                rng.integers(-lim, lim + 1, n - v_off, dtype=np.int64)])       # V elements: differences, up to the bound
        plane[v_off] = lim if rank == 0 else -lim                                              # (the extremes)
        plane[v_off + 1] = lim
        mx = torch.tensor([int(plane[:v_off].max()), int(np.abs(plane[v_off:]).max())], dtype=torch.int64)
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        width = pick_width(int(mx[0]), int(mx[1]), world)
        # the whole plane as one part, sent as `world` copies of it so that every rank receives the full sum
        words = _lanes16_pack(plane, v_off, world)
        send = torch.from_numpy(np.tile(words, world))
        recv = torch.empty(len(words), dtype=torch.int32)
        reduce_scatter_typed(send, recv, rank, world)
        got = _lanes16_unpack(recv.numpy(), n, v_off, world)
        np.savez(os.path.join(out_dir, "l%d.npz" % rank), plane=plane, got=got, width=width)
    finally:
        dist.destroy_process_group()


def test_sixteen_bit_lanes_add_up_inside_int32_sums(tmp_path):
    """The width-16 transport of the sharded finalize (no 16-bit integer type in RCCL): unsigned 16-bit lanes, two per int32 word,
    whose sums stay below 2^16 -- V elements biased by 32767 // world, E counts as four 8-bit digits -- summed as int32 by the
    collective (gloo here, world_size 2) and put back together give the exact 64-bit sums, at the bound of what pick_width
    admits; pick_width itself on the all-reduced maxima."""
    from bronko_amd.dist import pick_width
    world = 2
    mp.spawn(_worker_lanes, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "l%d.npz" % r)) for r in range(world)]
    total = got[0]["plane"] + got[1]["plane"]
    for r in range(world):
        assert np.array_equal(got[r]["got"], total), r
        assert int(got[r]["width"]) == 16
    assert pick_width(2 ** 32, 1, 2) == 64 and pick_width(10, 16384, 2) == 32 and pick_width(10, 16383, 2) == 16
    assert pick_width(2 ** 31, 5, 2) == 16 and pick_width(2 ** 30, 2 ** 30, 2) == 64 and pick_width(0, 0, 64) == 16
