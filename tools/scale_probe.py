#!/usr/bin/env python3
"""The reference's scale claim ("hundreds of strains against hundreds of samples", README.md:12) in numbers: N synthetic strains at
k = 31 -- index build and engine creation on the device, device memory of the tables and of every engine in flight, and a few
1 M-read samples through every genome's rows and through the selected genome's.   tools/scale_probe.py [strains=250] [samples=4]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import torch
from bronko_amd import Params, synth
from bronko_amd.engine import Engine, build_index_device, device_memory
n_strains = int(sys.argv[1]) if len(sys.argv) > 1 else 250
n_samples = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = torch.device("cuda", 0)
base = synth.read_fasta_bytes(os.path.join(ROOT, "tests", "golden", "4_sarscov2", "wuhan_ref.fasta"))
files = synth.strain_files(base, n_strains)
torch.cuda.synchronize()
free0 = device_memory(0)[0]
t0 = time.time()
built = build_index_device(31, files, device=0)
t1 = time.time()
eng = Engine(31, built[0], built[1], built[2], files, Params(device=0))
del built
t2 = time.time()
free1 = device_memory(0)[0]
print("%d strains, k = 31: index build %.2f s, engine create %.2f s; tables + one engine %.2f GB; counter plane %.2f GB, %d window slots" %
      (n_strains, t1 - t0, t2 - t1, (free0 - free1) / 1e9, eng.counter_len * 8 / 1e9, eng.n_slots), flush=True)
samples = []
for s in range(n_samples):
    g, isnv = synth.sample_genome(files[s % n_strains][1][0][1], 5 + s)
    codes = synth.single_end_codes_torch(g, 1000000, 150, 5 * 1000003 + s, err=0.005, isnv=isnv, device=dev)
    samples.append(synth.pack_codes_torch(codes))
    del codes
torch.cuda.synchronize()
for sel in (False, True):
    e0 = eng if not sel else eng.fork(Params(device=0, pileup_selected_only=True))
    f_before = device_memory(0)[0]
    engs = [e0] + [e0.fork() for _ in range(2)]
    per_fork = (f_before - device_memory(0)[0]) / 2
    def run(i):
        e = engs[i % len(engs)]
        w, l = samples[i % len(samples)]
        e.sample_begin(); e.push_reads_device(0, w.data_ptr(), w.shape[1], l.data_ptr(), l.numel()); e.sample_finalize(1)
    for i in range(2 * len(engs)): run(i)
    torch.cuda.synchronize()
    n = max(6, 2 * n_samples)
    t = time.time()
    for i in range(n): run(i)
    torch.cuda.synchronize()
    dt = (time.time() - t) / n
    res = engs[(n - 1) % len(engs)].sample_download(1, arrays=False)
    top = sorted([(int(v), i) for i, v in enumerate(res.stats[0, :, 0])], reverse=True)[:2]
    print("  %s: %.2f ms per 1 M-read sample, three in flight = %.0f M reads/s; one more engine in flight %.2f GB; best genomes %s" %
          ("selected genome's rows" if sel else "every genome's rows", dt * 1e3, 1.0 / dt, per_fork / 1e9, top), flush=True)
    for e in engs[1:]: e.close()
    if sel: e0.close()
print("device memory in use at the end: %.1f of %.1f GB" % ((device_memory(0)[1] - device_memory(0)[0]) / 1e9, device_memory(0)[1] / 1e9))
eng.close()
