# -*- coding:utf-8 -*-
"""Waveform-in embedding extractor - the command line of the reference's online script
(/root/reference/pytorch/pipeline/onestep/extract_embeddings_online.py:26-66; called by
pipeline/extract_xvectors_for_pytorch_new.sh):

    extract_embeddings_online.py [--nnet-config CFG | --model-blueprint PY --model-creation CTOR]
                                 --data-type raw --feat-config feat.yaml [--max-chunk N]
                                 [--use-gpu true] [--gpu-id N] model-path wav.scp vectors-wspecifier

wav.scp holds "utt-id path.wav" lines (reference libs/support/utils.py:625-638); feat.yaml holds the reference's
feature_extraction_conf: {feature_type: fbank|mfcc, kaldi_featset: {...torchaudio names...}, mean_var_conf: {...}}.

Where the reference decodes, computes torchaudio features, uploads and embeds one utterance at a time (lines 118-135),
this script ships 16-bit PCM (half the bytes of float features), and runs front-end + extractor on the device per batch:
    reader threads  wav files -> int16 samples packed in a pinned buffer
    device          H2D, asv_fbank_pcm16 (+ asv_cmvn), asv_net_extract, D2H
    writer          embeddings -> Kaldi vector ark (same bytes as kaldi_io.write_vec_flt per entry), input order
and prints "RTF:<device seconds per second of audio>" like the reference (line 138).

Not offered: --data-type shard (tar shards of the training pipeline), --de-silence (host-side waveform trimming of the
training egs, processor.py:149-175); --data-type kaldi never worked in the reference (egs_online.py:258-259 uses an undefined
dataset) and is rejected too.  WAV files must be 16-bit PCM; the first channel is used (processor.py:138).
"""

import argparse
import os
import sys
import traceback
import wave

sys.path.insert(0, "subtools/pytorch")
sys.path.insert(0, os.path.normpath(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")))

import numpy as np
import torch
import yaml

import libs.support.kaldi_io as kaldi_io
import libs.support.utils as utils
from libs.amd import frontend


def get_args(argv=None):
    parser = argparse.ArgumentParser(description="Extract embeddings form a piece of feats.scp or pipeline")
    parser.add_argument("--nnet-config", type=str, default="", help="This config contains model_blueprint and model_creation.")
    parser.add_argument("--model-blueprint", type=str, default=None, help="A *.py which includes the instance of nnet in this training.")
    parser.add_argument("--model-creation", type=str, default=None, help="A command to create the model class, e.g. Xvector(40,2).")
    parser.add_argument("--data-type", type=str, default="raw", choices=["raw", "shard", "kaldi"], help="raw: wav.scp")
    parser.add_argument("--de-silence", type=str, default="false", choices=["true", "false"], help="Not offered on this path.")
    parser.add_argument("--amp-th", type=int, default=50, help="De_silence threshold (16bit)")
    parser.add_argument("--max-chunk", type=int, default=10000, help="Select chunk_size of features when extracting xvector")
    parser.add_argument("--feat-config", type=str, default="", help="The config yaml of feat extraction")
    parser.add_argument("--use-gpu", type=str, default="true", choices=["true", "false"], help="The MI355X path needs a GPU; 'false' is rejected.")
    parser.add_argument("--gpu-id", type=str, default="", help="Specify a fixed gpu, or select gpu automatically.")
    parser.add_argument("--batch-seconds", type=float, default=1200.0, help="Upper bound of audio per batch.")
    parser.add_argument("--batch-utts", type=int, default=512, help="Upper bound of utterances per batch.")
    parser.add_argument("--num-readers", type=int, default=4, help="WAV reader threads.")
    parser.add_argument("model_path", metavar="model-path", type=str, help="The model used to extract embeddings.")
    parser.add_argument("feats_rspecifier", metavar="feats-rspecifier", type=str, help="wav.scp")
    parser.add_argument("vectors_wspecifier", metavar="vectors-wspecifier", type=str, help="")
    return parser.parse_args(argv)


def read_wav_pcm16(path):
    """(int16 samples of channel 0, sample rate).  16-bit PCM only - anything else is an error, not a silent conversion."""
    with wave.open(path, "rb") as f:
        if f.getsampwidth() != 2 or f.getcomptype() != "NONE":
            raise ValueError("%s: only 16-bit PCM WAV is supported (sample width %d, %s)" % (path, f.getsampwidth(), f.getcomptype()))
        ch, sr, n = f.getnchannels(), f.getframerate(), f.getnframes()
        data = np.frombuffer(f.readframes(n), dtype="<i2")
    return (data.reshape(-1, ch)[:, 0] if ch > 1 else data), sr


def extract_wavs(model, items, w, feat_conf, max_chunk, batch_seconds=1200.0, batch_utts=512, num_readers=4):
    """items: [(key, wav path)].  Returns (number written, device seconds, audio seconds as frames * 0.01 like the reference)."""
    import queue
    import threading
    from concurrent.futures import ThreadPoolExecutor
    engine = model._amd_engine()
    kind = feat_conf.get("feature_type", "mfcc")
    featset = dict(feat_conf.get("kaldi_featset", {}) or {})
    mv = feat_conf.get("mean_var_conf", {})
    mean_norm = bool(mv.get("mean_norm", True)) if mv is not None else False      # processor.py:410-413: {} -> defaults, None -> identity
    std_norm = bool(mv.get("std_norm", False)) if mv is not None else False
    dev = torch.device("cuda", engine.device_index)
    batches = queue.Queue(maxsize=2)

    free_bufs = queue.Queue()                    # two pinned staging buffers rotate between the producer and the consumer
    for _ in range(2):
        free_bufs.put(None)

    def produce():
        """Every file is read and decoded ONCE: decoded utterances that do not fit the audio budget of the batch being built
        stay in `pending` for the next one (the first version re-read the tail of every 512-file group 3-4 times)."""
        try:
            with ThreadPoolExecutor(max_workers=max(1, num_readers)) as pool:
                pending, pos = [], 0              # [(key, samples, rate)]
                while pos < len(items) or pending:
                    # decode ahead until the pending audio certainly fills one batch (or the list ends)
                    while pos < len(items) and (len(pending) < batch_utts and sum(len(wv) / float(sr) for _, wv, sr in pending) <= batch_seconds):
                        group = items[pos:pos + max(8, num_readers * 4)]
                        for (key, _), (wav, sr) in zip(group, pool.map(lambda it: read_wav_pcm16(it[1]), group)):
                            pending.append((key, wav, sr))
                        pos += len(group)
                    # cut at the audio budget / utterance cap (at least one utterance)
                    total, n = 0.0, 0
                    for _, wav, sr in pending[:batch_utts]:
                        if n > 0 and total + len(wav) / float(sr) > batch_seconds:
                            break
                        total += len(wav) / float(sr)
                        n += 1
                    take, pending = pending[:n], pending[n:]
                    rates = {sr for _, _, sr in take}
                    if len(rates) != 1:
                        raise ValueError("mixed sample rates in one batch: %s" % sorted(rates))
                    off = np.zeros(n + 1, dtype=np.int64)
                    np.cumsum([len(wav) for _, wav, _ in take], out=off[1:])
                    host = free_bufs.get()        # blocks until the consumer has uploaded the batch that used it
                    if host is None or host.numel() < int(off[-1]):
                        host = torch.empty(max(int(off[-1]), int(batch_seconds * 16000)), dtype=torch.int16).pin_memory()
                    hv = host.numpy()
                    for (_, wav, _), a, b in zip(take, off[:-1], off[1:]):
                        hv[a:b] = wav
                    batches.put(([k for k, _, _ in take], host, off, rates.pop()))
            batches.put(None)
        except BaseException as e:
            batches.put(e)

    t = threading.Thread(target=produce, daemon=True)
    t.start()
    n_done, dev_ms, audio_s = 0, 0.0, 0.0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.device(dev):
        while True:
            item = batches.get()
            if item is None:
                break
            if isinstance(item, BaseException):
                raise item
            keys, host, off, sr = item
            wave_dev = host[:int(off[-1])].to(dev, non_blocking=True)
            torch.cuda.current_stream(dev).synchronize()             # the upload has read the pinned buffer: hand it back
            free_bufs.put(host)
            e0.record()
            feats, frame_off = frontend.fbank_device(wave_dev, off, kind=kind, mean_norm=mean_norm, std_norm=std_norm,
                                                     **dict(featset, sample_frequency=float(sr)))    # processor.py:428
            keep = [i for i in range(len(keys)) if frame_off[i + 1] > frame_off[i]]
            if len(keep) != len(keys):
                for i in set(range(len(keys))) - set(keep):
                    print("Warning: {0} is shorter than one frame, skipped".format(keys[i]))
            emb = _extract_subset(engine, feats, frame_off, keep, max_chunk) if keep else None
            e1.record()
            if emb is not None:
                out = emb.cpu().numpy()
                e1.synchronize()
                dev_ms += e0.elapsed_time(e1)
                audio_s += float(frame_off[-1]) * 0.01
                w.write(kaldi_io.vec_flt_ark_bytes([keys[i] for i in keep], out))
                n_done += len(keep)
    t.join()
    return n_done, dev_ms * 1e-3, audio_s


def _extract_subset(engine, feats, frame_off, keep, max_chunk):
    """Utterances with at least one frame only (the others have no rows in `feats`, so the packed matrix is unchanged)."""
    lens = [int(frame_off[i + 1] - frame_off[i]) for i in keep]
    offs = np.zeros(len(keep) + 1, dtype=np.int32)
    np.cumsum(lens, out=offs[1:])
    return engine.extract_device_guarded(feats, offs, max_chunk=max_chunk)       # (synchronous loop: the range guard's wait costs nothing here)


def main(argv=None):
    print(" ".join(sys.argv))
    args = get_args(argv)
    try:
        if args.nnet_config != "":
            model_blueprint, model_creation = utils.read_nnet_config(args.nnet_config)
        elif args.model_blueprint is not None and args.model_creation is not None:
            model_blueprint, model_creation = args.model_blueprint, args.model_creation
        else:
            raise ValueError("Expected nnet_config or (model_blueprint, model_creation) to exist.")
        if not utils.to_bool(args.use_gpu):
            raise RuntimeError("asv-subtools_amd extracts on a ROCm device only (--use-gpu=true); there is no CPU path")
        if args.data_type != "raw":
            raise ValueError("Do not support datatype: {0} now.".format(args.data_type))
        if utils.to_bool(args.de_silence):
            raise ValueError("--de-silence is not offered by the device path")
        with open(args.feat_config, "r") as fin:
            feat_conf = yaml.load(fin, Loader=yaml.FullLoader) or {}

        model = utils.create_model_from_py(model_blueprint, model_creation)
        model.load_state_dict(torch.load(args.model_path, map_location="cpu"), strict=False)
        model = utils.select_model_device(model, args.use_gpu, gpu_id=args.gpu_id)
        model.eval()
        items = []
        with open(args.feats_rspecifier, "r", encoding="utf8") as fin:
            for line in fin:
                arr = line.strip().split()
                if not arr:
                    continue
                assert len(arr) == 2, "wav.scp lines are 'utt-id path.wav' (pipes are not supported): %r" % line
                items.append((arr[0], arr[1]))
        with kaldi_io.open_or_fd(args.vectors_wspecifier, "wb") as w:
            n_done, dev_s, audio_s = extract_wavs(model, items, w, feat_conf, args.max_chunk, args.batch_seconds, args.batch_utts, args.num_readers)
        print("Extracted {0} embeddings.".format(n_done))
        print("RTF:{:.7f}".format(dev_s / audio_s if audio_s > 0 else 0.0))
    except BaseException as e:
        if not isinstance(e, KeyboardInterrupt):
            traceback.print_exc()
        sys.exit(1)


if __name__ == "__main__":
    main()
