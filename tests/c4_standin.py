#!/usr/bin/env python3
"""BASELINE configs[3] stand-in at the size SURVEY.md 8(d) defines (VoxCeleb1-O shapes: 4 708 utterances, T ~ U(400, 1500),
37 720 trials with 50 % targets from planted speakers): ECAPA-TDNN extraction sharded over the GPUs of one node, ONE RCCL
all-gather of the embeddings, cosine scoring (sub-mean + length-norm + dot products) and EER on rank 0 - and the same chain
on reference-equivalent embeddings (exact-f32 extraction, tied to the numpy oracle on sampled utterances) for the delta.

    python tests/c4_standin.py [--gpus N] [--precision bf16|f32x] [--model ecapa|xvector] [--utts 4708] ...
    python -m torch.distributed.run --nproc-per-node N ... tests/c4_standin.py --gpus N ...

With --gpus N > 1 and no launcher environment the script spawns its own ranks (as bench.py does).  Prints one JSON line on
rank 0.  Lives under tests/ because it uses the oracle as its checker; the product pieces it drives are
libs.amd.shard (partition + all-gather), the engine and libs.amd.scoring.  `--fake-extractor` replaces the engine by a
deterministic numpy stand-in and the device scoring by the numpy oracle so that the whole control flow runs under gloo on
CPU (tests/test_c4_standin_gloo.py)."""

import argparse
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(REPO, "asv-subtools_amd", "pytorch"), REPO, os.path.join(REPO, "tests")]

import numpy as np


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--model", default="ecapa", choices=["ecapa", "xvector"])
    ap.add_argument("--precision", default="bf16", help="bf16 | f16 | f32x[-bf16|-f16] | f32")
    ap.add_argument("--utts", type=int, default=4708)
    ap.add_argument("--per-spk", type=int, default=4)
    ap.add_argument("--t-lo", type=int, default=400)
    ap.add_argument("--t-hi", type=int, default=1500)
    ap.add_argument("--trials", type=int, default=37720)
    ap.add_argument("--noise", type=float, default=0.8)
    ap.add_argument("--batch-frames", type=int, default=200_000)
    ap.add_argument("--oracle-checks", type=int, default=3, help="utterances compared with the numpy oracle (f32 pass)")
    ap.add_argument("--fake-extractor", action="store_true", help="CPU control-flow run (gloo): numpy stand-in for the engine and the device scoring")
    return ap.parse_args(argv)


class PlantedSet(object):
    """Utterance i = speaker pattern (a fixed random feature track per speaker) + noise, generated on demand from seeds, so
    every rank can produce exactly its shard without any file."""

    def __init__(self, n_utts, per_spk, t_lo, t_hi, dim, noise, seed=17):
        from libs.amd import synth
        self.n, self.dim, self.noise, self.t_hi, self.seed = n_utts, dim, noise, t_hi, seed
        self.labels = np.arange(n_utts) // per_spk
        self.lengths = synth.synth_lengths(n_utts, t_lo, t_hi, seed)
        self._base = {}

    def utt(self, i):
        from libs.amd import synth
        spk = int(self.labels[i])
        if spk not in self._base:
            if len(self._base) > 8:
                self._base.clear()
            self._base[spk] = synth.synth_feats(self.t_hi, self.dim, 700_000 + spk)
        T = int(self.lengths[i])
        r = np.random.RandomState(900_000 + self.seed * 7919 + i)
        return (self._base[spk][:T] + self.noise * r.standard_normal((T, self.dim)).astype(np.float32)).astype(np.float32)


def fake_embedding(mat, dim=24):
    v = np.zeros(dim, dtype=np.float32)
    v[:dim - 1] = mat[:, :dim - 1].mean(axis=0) + 0.05 * mat[: 50, :dim - 1].std(axis=0)
    v[-1] = 1e-3 * mat.shape[0]
    return v


def cosine_eer(emb, ei, ti, tgt, fake):
    """sub-mean -> length-norm -> dot products -> EER (score/process.sh:177-203, score/score.sh:82-97, computeEER)"""
    if fake:
        from oracle import scoring_oracle as S
        x = S.length_normalize(emb, S.global_mean(emb))
        scores = S.dot_trials(x, x, ei, ti)
        return 100.0 * S.compute_eer(scores, tgt)[0], scores
    from libs.amd import scoring
    e = scoring.length_normalize(emb, scoring.mean_vector(emb))
    scores = scoring.score_trials(e, e, ei, ti)
    return scoring.eer(scores, tgt)[0], scores.cpu().numpy()


def run(args):
    import torch
    import torch.distributed as dist
    from libs.amd import shard, synth
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    fake = args.fake_extractor
    dev = None
    if not fake:
        assert torch.cuda.is_available(), "c4_standin.py needs a ROCm device (or --fake-extractor)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if fake:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=dev)
    data = PlantedSet(args.utts, args.per_spk, args.t_lo, args.t_hi, 80, args.noise)
    ei, ti, tgt = synth.synth_trials(data.labels, args.trials, seed=43)

    sd = None
    if not fake:
        import helpers
        blueprint, creation = {"ecapa": ("ecapa_tdnn_xvector.py", "ECAPA_TDNN(80,10,training=False)"),
                               "xvector": ("xvector.py", "Xvector(80,10,training=False)")}[args.model]
        model = helpers.build_model(blueprint, creation)
        sd = synth.synth_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, 0)
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model.cuda()

    def extract_all(precision):
        if fake:
            extract_batch = lambda mats: torch.from_numpy(np.stack([fake_embedding(m) for m in mats]))
        else:
            model.amd_precision = precision
            eng = model._amd_engine()

            def extract_batch(mats):
                offs = np.zeros(len(mats) + 1, dtype=np.int32)
                np.cumsum([m.shape[0] for m in mats], out=offs[1:])
                return eng.extract_device(torch.from_numpy(np.concatenate(mats, axis=0)).to(dev), offs)
        t0 = time.perf_counter()
        emb = shard.extract_sharded(extract_batch, data.lengths, data.utt, max_frames=args.batch_frames, max_utts=1024, device=dev)
        if dev is not None:
            torch.cuda.synchronize(dev)
        return emb, time.perf_counter() - t0

    emb_new, secs_new = extract_all(args.precision)
    res = None
    if fake:
        emb_ref = emb_new
    else:
        emb_ref, _ = extract_all("f32")                 # reference-equivalent embeddings (checked against the oracle below)
    if rank == 0:
        eer_new, s_new = cosine_eer(emb_new, ei, ti, tgt, fake)
        eer_ref, s_ref = cosine_eer(emb_ref, ei, ti, tgt, fake)
        res = {"config": "BASELINE configs[3] stand-in: %s, %d planted-speaker utterances, T ~ U(%d, %d), %d trials" % (args.model, args.utts, args.t_lo, args.t_hi, args.trials),
               "n_gpus": world, "precision": args.precision, "frames": int(data.lengths.sum()),
               "eer_percent": round(float(eer_new), 4), "eer_reference_equivalent_percent": round(float(eer_ref), 4),
               "eer_delta_percent": round(float(eer_new - eer_ref), 4), "max_abs_score_delta": float("%.3g" % np.abs(s_new - s_ref).max()),
               "extract_seconds_incl_host_generation": round(secs_new, 2), "all_gather": "one all_gather_into_tensor of [n_pad, E] f32 per extraction"}
        if not fake and args.oracle_checks > 0:
            from oracle import np_oracle as O
            fn = (lambda c: O.ecapa_embed(c, sd, "near")) if args.model == "ecapa" else (lambda c: O.xvector_embed(c, sd, "far"))
            pos = list(np.argsort(data.lengths, kind="stable")[:args.oracle_checks])       # the shortest ones: seconds each on the CPU
            want = np.stack([O.extract_embedding(fn, data.utt(i)) for i in pos])
            got = emb_ref.cpu().numpy()[pos]
            res["oracle_max_rel_err_f32"] = float("%.3g" % (np.abs(got - want).max() / np.abs(want).max()))
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return res, emb_new


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        sys.exit(subprocess.call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
                                  "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]))
    run(args)


if __name__ == "__main__":
    main()
