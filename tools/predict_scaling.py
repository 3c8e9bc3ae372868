"""Multi-GPU pre-flight (no multi-GPU node is reachable from the build sessions): what `bench.py --gpus N` should print
on an 8-GPU node, computed on ONE GPU from the only thing that differs between ranks - the problems they fit.  Rank r of a weak-
scaling run fits the global frames [32 r, 32 r + 32) (bench.py: seeds follow the global frame index): this script fits exactly
those shards one after the other on the one GPU it has, times each (a rank's step = its shard's complete fit) and predicts
    ms_per_step(N) = max over ranks r < N of shard_ms[r]          (barrier + max-over-ranks timing of the contract)
    value(N)       = sum over ranks r < N of shard_closures[r] / ms_per_step(N)
i.e. weak scaling with the spread of the slowest problem per rank as the only loss (no data-path collective; the final all_gather
of [32, 118] floats per rank is ~15 KB - microseconds over xGMI).  Strong scaling (--strong, a fixed total split over the ranks) is
bounded by the round latency: a shard's fit lasts (closure rounds of its slowest problem) x (time per round), whatever its size.
    python tools/predict_scaling.py [configs1|configs3] [out.json]"""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import bench
    from mvsmplfitting_amd import synthetic as syn
    from mvsmplfitting_amd.engine import MvFit, stage_weights
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'configs1'
    persons = 4 if cfg == 'configs3' else 1
    model = syn.make_body_model(0, skin_topk=4)
    eng = MvFit(model)
    stages = stage_weights(1536.0)
    shards = []
    for r in range(8):
        cams, gt, conf, x0 = bench.build_inputs(eng, syn, 32 * r, 32 * r + 32, persons, 8)
        x0_d = torch.tensor(x0, device=eng.device)
        eng.fit(x0_d, stages)
        torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            xf, st = eng.fit(x0_d, stages)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        ncl = st['n_closure'].cpu().numpy()
        shards.append(dict(rank=r, frames=[32 * r, 32 * r + 32], problems=int(ncl.shape[0]), ms=round(1e3 * float(np.median(ts)), 3),
                           closures=int(ncl.sum()), closure_rounds=int(ncl.max()), closures_mean=round(float(ncl.mean()), 1),
                           us_per_round=round(1e6 * float(np.median(ts)) / int(ncl.max()), 2), resident_form=eng.pass_profile()['form']))
    eng.close()
    pred = []
    for n in (1, 2, 4, 8):
        ms = max(s['ms'] for s in shards[:n])
        cl = sum(s['closures'] for s in shards[:n])
        pred.append(dict(n_gpus=n, ms_per_step=ms, value=round(cl / ms * 1e3, 1),
                         efficiency_vs_n_times_rank0=round(cl / ms / (n * shards[0]['closures'] / shards[0]['ms']), 4),
                         slowest_rank=int(np.argmax([s['ms'] for s in shards[:n]]))))
    out = dict(workload=cfg, note=__doc__.split('\n\n')[0].replace('\n', ' '), shards=shards, predicted_weak_scaling=pred,
               strong_scaling_bound='ms_per_step >= closure rounds of the slowest problem x us_per_round, whatever N: %d rounds x %.1f us = %.2f ms for the frames of rank 0'
                                    % (shards[0]['closure_rounds'], shards[0]['us_per_round'], shards[0]['closure_rounds'] * shards[0]['us_per_round'] * 1e-3))
    print(json.dumps(out, indent=1))
    if len(sys.argv) > 2:
        with open(sys.argv[2], 'w') as f:
            json.dump(out, f, indent=1)


if __name__ == '__main__':
    main()
