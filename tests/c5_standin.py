#!/usr/bin/env python3
"""BASELINE configs[4] stand-in at its stated shape: "ResNet34-SE (2D fbank) extractor + PLDA back-end (score/pyplda),
variable-length 200-1000 frame utts, 8 x MI355X" (SURVEY.md 8(d) C5).

    * ResNetXvector(80, ..., use_se, original BasicBlock form, fc2 without non-linearity: the launcher's configuration,
      runResnetXvector_online.py:221-260) on >= 2 000 planted-speaker utterances of T ~ U[200, 1000] frames, PACKED RAGGED in
      length-sorted batches - every stride-2 stage keeps its own L_out = floor((L - 1) / 2) + 1 per utterance, the per-bin
      pooling runs over true lengths (reference: model/resnet_xvector.py:183-208, batch = 1);
    * the utterances shard over the GPUs of the node by length, ONE all-gather collects the embeddings (libs.amd.shard);
    * PLDA back-end on rank 0: 10 EM iterations (score/pyplda/plda_base.py:248-300) on the embeddings of the training
      speakers (device, float64), simultaneous diagonalisation, transform + length normalisation, log-likelihood ratios of
      the evaluation trials (plda_base.py:93-136), EER;
    * the same chain - PLDA training included - on reference-equivalent embeddings (the exact-f32 extraction, tied to the
      numpy oracle on sampled utterances) gives the EER delta of the precision mode under test; a sample of the trials is
      re-scored by the float64 oracle (oracle/scoring_oracle.py) with the same PLDA parameters.

    python tests/c5_standin.py [--gpus N] [--precision bf16|f16|f32x|f32] [--utts 2000] ...

Prints one JSON line on rank 0.  `--fake-extractor` replaces the engine and the device PLDA by numpy stand-ins so that the
control flow runs on gloo CPU ranks (tests/test_sharded_script_gloo.py).  Lives under tests/: the oracle is its checker."""

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

CREATION = ("ResNetXvector(80,10,training=False,resnet_params={'use_se':True,'se_ratio':4,'full_pre_activation':False},"
            "fc2_params={'nonlinearity':'','bn_params':{'momentum':0.5,'affine':False,'track_running_stats':True}})")


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--precision", default="f32x")
    ap.add_argument("--utts", type=int, default=2000)
    ap.add_argument("--per-spk", type=int, default=4)
    ap.add_argument("--train-frac", type=float, default=0.6, help="share of the speakers whose embeddings train the PLDA")
    ap.add_argument("--t-lo", type=int, default=200)
    ap.add_argument("--t-hi", type=int, default=1000)
    ap.add_argument("--trials", type=int, default=20000)
    ap.add_argument("--noise", type=float, default=0.8)
    ap.add_argument("--plda-iters", type=int, default=10)
    ap.add_argument("--batch-frames", type=int, default=120_000)
    ap.add_argument("--oracle-checks", type=int, default=3, help="utterances compared with the numpy oracle (f32 pass) + 200 oracle-scored trials")
    ap.add_argument("--fake-extractor", action="store_true", help="CPU control-flow run (gloo): numpy stand-ins for the engine and the device PLDA")
    return ap.parse_args(argv)


def fake_embedding(mat, dim=24):
    v = np.zeros(dim, dtype=np.float32)
    v[:dim - 1] = mat[:, :dim - 1].mean(axis=0) + 0.05 * mat[: 50, :dim - 1].std(axis=0)
    v[-1] = 1e-3 * mat.shape[0]
    return v


def plda_chain(emb, labels, train_mask, ei, ti, tgt, iters, fake):
    """PLDA training on the training speakers' embeddings, LLR of the evaluation trials, EER.  emb: [n, E] (tensor or array);
    ei / ti index the evaluation utterances (positions inside emb[~train_mask])."""
    x = emb.cpu().numpy() if hasattr(emb, "cpu") else np.asarray(emb)
    tr_x, tr_l = x[train_mask], labels[train_mask]
    ev = x[~train_mask]
    if fake:
        from oracle import scoring_oracle as S
        stats = S.PldaStats(tr_x.shape[1])
        for spk in np.unique(tr_l):
            stats.add_samples(1.0, tr_x[tr_l == spk].astype(np.float64))
        mean, within, between = S.plda_em(stats, num_iters=iters)
        tr1, psi = S.plda_diagonalise(within, between)
        t = np.stack([S.plda_transform(v.astype(np.float64), mean, tr1, psi, 1) for v in ev])
        llr = np.array([S.plda_llr(t[a], 1, t[b], psi) for a, b in zip(ei, ti)])
        return 100.0 * S.compute_eer(llr, tgt)[0], llr, (mean, tr1, psi)
    from libs.amd import scoring
    mean, within, between = scoring.train_plda(tr_x, tr_l, num_iters=iters)
    plda = scoring.Plda.from_covariances(mean, within, between)
    t_dev = plda.transform_vectors(ev)
    llr = plda.llr_trials(t_dev, t_dev, ei, ti)
    return scoring.eer(llr, tgt)[0], llr.cpu().numpy(), (plda.mean, plda.transform, plda.psi)


def run(args):
    import torch
    import torch.distributed as dist
    from libs.amd import shard, synth
    from c4_standin import PlantedSet
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    fake = args.fake_extractor
    dev = None
    if not fake:
        assert torch.cuda.is_available(), "c5_standin.py needs a ROCm device (or --fake-extractor)"
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo") if fake else dist.init_process_group(backend="nccl", device_id=dev)
    data = PlantedSet(args.utts, args.per_spk, args.t_lo, args.t_hi, 80, args.noise, seed=29)
    n_spk = int(data.labels.max()) + 1
    train_mask = data.labels < int(round(args.train_frac * n_spk))
    ei, ti, tgt = synth.synth_trials(data.labels[~train_mask], args.trials, seed=47)

    sd = None
    if not fake:
        import helpers
        model = helpers.build_model("resnet_xvector.py", CREATION)
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
        emb = shard.extract_sharded(extract_batch, data.lengths, data.utt, max_frames=args.batch_frames, max_utts=512, device=dev)
        if dev is not None:
            torch.cuda.synchronize(dev)
        return emb, time.perf_counter() - t0

    emb_new, secs_new = extract_all(args.precision)
    emb_ref = emb_new if fake else extract_all("f32")[0]            # reference-equivalent embeddings (checked against the oracle below)
    res = None
    if rank == 0:
        eer_new, llr_new, _ = plda_chain(emb_new, data.labels, train_mask, ei, ti, tgt, args.plda_iters, fake)
        eer_ref, llr_ref, (mean, transform, psi) = plda_chain(emb_ref, data.labels, train_mask, ei, ti, tgt, args.plda_iters, fake)
        got, want = (emb_new.cpu().numpy(), emb_ref.cpu().numpy()) if not fake else (emb_new.numpy(), emb_ref.numpy())
        res = {"config": "BASELINE configs[4] stand-in: ResNet34-SE + PLDA, %d planted-speaker utterances (%d speakers, %d %% train the PLDA), T ~ U(%d, %d), "
                         "%d EM iterations, %d LLR trials" % (args.utts, n_spk, round(100 * args.train_frac), args.t_lo, args.t_hi, args.plda_iters, args.trials),
               "n_gpus": world, "precision": args.precision, "frames": int(data.lengths.sum()),
               "eer_percent": round(float(eer_new), 4), "eer_reference_equivalent_percent": round(float(eer_ref), 4),
               "eer_delta_percent": round(float(eer_new - eer_ref), 4), "max_abs_llr_delta": float("%.3g" % np.abs(llr_new - llr_ref).max()),
               "embedding_max_rel_err_vs_f32": float("%.3g" % (np.abs(got - want).max() / np.abs(want).max())),
               "extract_seconds_incl_host_generation": round(secs_new, 2), "all_gather": "one all_gather_into_tensor of [n_pad, E] f32 per extraction"}
        if not fake and args.oracle_checks > 0:
            from oracle import np_oracle as O
            from oracle import scoring_oracle as S
            pos = list(np.argsort(data.lengths, kind="stable")[:args.oracle_checks]) + [int(np.argmax(data.lengths))]      # the shortest ones + the longest
            owant = np.stack([O.extract_embedding(lambda c: O.resnet_embed(c, sd, "near", ""), data.utt(i)) for i in pos])
            res["oracle_max_rel_err_f32"] = float("%.3g" % (np.abs(want[pos] - owant).max() / np.abs(owant).max()))
            ev = want[~train_mask]
            t_ref = {int(k): S.plda_transform(ev[int(k)].astype(np.float64), mean, transform, psi, 1) for k in set(ei[:200]) | set(ti[:200])}
            llr_o = np.array([S.plda_llr(t_ref[int(a)], 1, t_ref[int(b)], psi) for a, b in zip(ei[:200], ti[:200])])
            res["oracle_max_abs_llr_err_200_trials"] = float("%.3g" % np.abs(llr_ref[:200] - llr_o).max())
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
