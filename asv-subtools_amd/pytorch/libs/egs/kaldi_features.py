# -*- coding:utf-8 -*-
"""Online feature extraction with the reference's interface (pytorch/libs/egs/kaldi_features.py:70-137):
KaldiFeature(feature_type, kaldi_featset, mean_var_conf)(waveforms, lengths) -> list of [frames, dim] matrices.
The whole batch is one asv_fbank launch on the MI355X (libs/amd/frontend.py) instead of a torchaudio call per utterance;
the matrices come back as CUDA tensors ready for extract_embedding_batch."""

from libs.amd import frontend


class InputSequenceNormalization(object):
    """Per-utterance mean / std normalisation of a [t, f] matrix (reference kaldi_features.py:11-66).  Used through
    KaldiFeature it is fused into the device call; called directly it runs asv_cmvn on one matrix."""

    def __init__(self, mean_norm=True, std_norm=False):
        self.mean_norm, self.std_norm, self.eps = mean_norm, std_norm, 1e-10

    def __call__(self, x):
        import ctypes as C
        import numpy as np
        import torch
        from libs.amd import capi
        x = (x if isinstance(x, torch.Tensor) else torch.from_numpy(np.asarray(x))).to(torch.float32)
        x = (x if x.is_cuda else x.cuda()).contiguous().clone()
        off = np.array([0, x.shape[0]], dtype=np.int64)
        with torch.cuda.device(x.device):
            capi.check(capi.lib().asv_cmvn(C.c_void_p(x.data_ptr()), off.ctypes.data_as(C.POINTER(C.c_longlong)), 1, x.shape[1],
                                           int(self.mean_norm), int(self.std_norm), self.eps,
                                           C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)), "asv_cmvn")
        return x


class KaldiFeature(object):
    def __init__(self, feature_type='mfcc', kaldi_featset={}, mean_var_conf={}):
        assert feature_type in ['mfcc', 'fbank']
        self.feat_type = feature_type
        self.kaldi_featset = dict(kaldi_featset)
        self.mean_var_conf = dict(mean_var_conf)
        frontend.fbank_options(feature_type, **self.kaldi_featset)        # option errors at construction, like a bad config should

    def __call__(self, waveforms, lengths=None):
        """waveforms: [batch, time] (or [batch, time, 1]) tensor, lengths: relative lengths [batch] or None."""
        import torch
        if torch.any(torch.isnan(waveforms)):
            raise ValueError('feats:{}'.format(waveforms))
        if waveforms.dim() == 3:
            if waveforms.shape[2] != 1:
                raise ValueError("KaldiFeature: one channel per utterance")
            waveforms = waveforms[:, :, 0]
        waves = []
        for i, wav in enumerate(waveforms):
            n = int((lengths[i] * waveforms.shape[1]).long()) if lengths is not None else wav.shape[0]
            waves.append(wav[:n])
        mv = self.mean_var_conf
        return frontend.fbank(waves, mean_norm=bool(mv.get('mean_norm', True)) if mv else False,
                              std_norm=bool(mv.get('std_norm', False)) if mv else False, kind=self.feat_type, **self.kaldi_featset)
