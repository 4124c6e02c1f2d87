"""One sample's reads sharded over ranks, end to end through bronko_amd.dist on real process groups: two processes share the
test box's one GPU (gloo between them; on a multi-GPU node the same code runs over RCCL), each scans its half of the reads,
ShardedFinalize exchanges the statistics tables (all-to-all), reduce-scatters the counter planes, maps its part and combines
the results.  Every rank must end with the oracle's pileup, statistics and -- full_kmer_stats -- KMC's distinct / counted totals."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from bronko_amd import synth
from tests import helpers

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _rank(rank, world, port, out_dir, sars_paths, kmer_stats):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bronko_amd import Params, pack_reads
        from bronko_amd.dist import ShardedFinalize, shard_bounds
        from bronko_amd.hostlib import HostIndex
        dev = torch.device("cuda", 0)
        ix = HostIndex.build(21, sars_paths, threads=2)
        eng = ix.engine(Params(full_kmer_stats=kmer_stats, kmer_table_log2=16))
        gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[2]), 23)
        c1, c2 = synth.paired_codes(gm, 12000, 150, 23, isnv=isnv)
        mates = [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)]
        stream = torch.cuda.ExternalStream(eng.stream_ptr(), device=dev)
        with torch.cuda.stream(stream):
            fin = ShardedFinalize(eng, 2, rank, world, dev)
            for rep in range(2):                                       # the engine is reusable: a second sample gives the same
                eng.sample_begin()
                for m, reads in enumerate(mates):
                    lo, hi = shard_bounds(len(reads), rank, world)
                    w, l = pack_reads(reads[lo:hi], 21)
                    eng.push_reads(m, w, l)
                fin()
                res = eng.sample_download(2)
        np.savez(os.path.join(out_dir, "r%d.npz" % rank), fwd_depth=res.fwd_depth, rev_depth=res.rev_depth, fwd_nk=res.fwd_nk, rev_nk=res.rev_nk,
                 stats=res.stats, present=res.present, kmer_stats=res.kmer_stats)
        eng.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kmer_stats", [False, True])
def test_two_ranks_share_one_samples_reads(oracle, sars_paths, tmp_path, kmer_stats):
    world = 2
    mp.spawn(_rank, args=(world, _free_port(), str(tmp_path), list(sars_paths), kmer_stats), nprocs=world, join=True)
    ix = oracle.Index.build(21, sars_paths)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[2]), 23)
    c1, c2 = synth.paired_codes(gm, 12000, 150, 23, isnv=isnv)
    pile = oracle.sample_pileup(ix, [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)])
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk"):
            assert np.array_equal(got[name], getattr(pile, name)), (r, name)
        assert np.array_equal(got["stats"], pile.stats) and np.array_equal(got["present"], pile.present)
        assert got["kmer_stats"][:, 1].tolist() == pile.kmc_stats[:, 1].tolist()
        if kmer_stats:
            assert got["kmer_stats"][:, 2:4].tolist() == pile.kmc_stats[:, 2:4].tolist(), (r, got["kmer_stats"], pile.kmc_stats)
    ix.close()


def _one_rank_over_rccl(rank, world, port, out_dir, sars_paths):
    """One rank, backend nccl (= RCCL): every collective of bronko_amd/dist.py is forced through the backend although there is
    nobody to talk to (force_collectives) -- reduce_scatter_tensor on the engine's packed planes at every width, the three small
    all-reduces, the two-word all-reduce(max) of "auto", the all-to-all of the statistics tables, the plain all-reduce of a u64
    plane -- all on the engine's ExternalStream, results against the oracle in the parent."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from bronko_amd import Params, pack_reads
        from bronko_amd.dist import DeviceVector, ShardedFinalize, allreduce_counters
        from bronko_amd.hostlib import HostIndex
        assert dist.get_backend() == "nccl"
        ix = HostIndex.build(21, sars_paths, threads=2)
        eng = ix.engine(Params(full_kmer_stats=True, kmer_table_log2=16))
        gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[2]), 23)
        c1, c2 = synth.paired_codes(gm, 12000, 150, 23, isnv=isnv)
        packed = [pack_reads(synth.codes_to_ascii(c), 21) for c in (c1, c2)]
        stream = torch.cuda.ExternalStream(eng.stream_ptr(), device=dev)
        out = {}
        with torch.cuda.stream(stream):
            for width in (16, 32, 64, "auto"):
                fin = ShardedFinalize(eng, 2, 0, 1, dev, width=width, force_collectives=True, time_comm=True)
                eng.sample_begin()
                for m, (w, l) in enumerate(packed):
                    eng.push_reads(m, w, l)
                fin()
                res = eng.sample_download(2)
                rs, cb, n = fin.comm_ms()
                assert n == 1 and rs > 0 and cb > 0 and fin.bytes_sent > 0
                assert not eng.transport_overflow()
                for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk", "stats", "present", "kmer_stats"):
                    out["%s_%s" % (name, width)] = getattr(res, name)
                out["widths_%s" % width] = np.array(fin.last_widths)
            # the plain form: ONE all-reduce(sum) of the u64 plane per mate file, then the ordinary finalize
            eng.sample_begin()
            for m, (w, l) in enumerate(packed):
                eng.push_reads(m, w, l)
            for m in range(2):
                plane = torch.as_tensor(DeviceVector(eng.counters_ptr(m), eng.counter_len), device=dev)
                before = plane.clone()
                allreduce_counters(plane, force=True)
                assert torch.equal(plane, before)
            res = eng.sample_finish(2)
            for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk", "stats", "present", "kmer_stats"):
                out["%s_allreduce" % name] = getattr(res, name)
        np.savez(os.path.join(out_dir, "nccl.npz"), **out)
        eng.close()
    finally:
        dist.destroy_process_group()


def test_one_rank_through_rccl(oracle, sars_paths, tmp_path):
    """First contact with RCCL on the one-GPU test box: a process group of one rank with backend `nccl`, every collective of
    bronko_amd/dist.py issued for real (world_size 1 is no shortcut inside the backend) on the engine's stream."""
    mp.spawn(_one_rank_over_rccl, args=(1, _free_port(), str(tmp_path), list(sars_paths)), nprocs=1, join=True)
    ix = oracle.Index.build(21, sars_paths)
    gm, isnv = synth.sample_genome(synth.read_fasta_bytes(sars_paths[2]), 23)
    c1, c2 = synth.paired_codes(gm, 12000, 150, 23, isnv=isnv)
    pile = oracle.sample_pileup(ix, [synth.codes_to_ascii(c1), synth.codes_to_ascii(c2)])
    got = np.load(os.path.join(str(tmp_path), "nccl.npz"))
    for tag in ("16", "32", "64", "auto", "allreduce"):
        for name in ("fwd_depth", "rev_depth", "fwd_nk", "rev_nk", "stats", "present"):
            assert np.array_equal(got["%s_%s" % (name, tag)], getattr(pile, name)), (tag, name)
        assert got["kmer_stats_%s" % tag][:, 1:4].tolist() == pile.kmc_stats[:, 1:4].tolist(), tag
    assert got["widths_auto"].tolist() == [16, 16]
    ix.close()
