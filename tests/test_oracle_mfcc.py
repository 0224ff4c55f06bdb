"""MFCC oracle (oracle/mfcc_np.py).  torchaudio is absent here and the reference holds no
vectors for it, so parity with the dependency is UNPINNED; what can be pinned is the
STFT/power stage against torch.stft (the primitive torchaudio.transforms.Spectrogram calls)
and the textbook properties of the mel/DCT matrices."""
import numpy as np
import torch

from oracle import mfcc_np


def test_power_spectrogram_matches_torch_stft():
    clips = mfcc_np.synth_clips(3, seed=11)
    for c in clips:
        x = torch.from_numpy((c / 32768.0).astype(np.float32))
        st = torch.stft(x, n_fft=512, hop_length=160, win_length=400,
                        window=torch.hamming_window(400), center=True, pad_mode='reflect',
                        normalized=False, onesided=True, return_complex=True)
        ref = st.abs().pow(2.0).numpy().T                     # (T, 257)
        got = mfcc_np.power_spectrogram((c / 32768.0).astype(np.float32))
        assert got.shape == ref.shape == (101, 257)
        assert np.max(np.abs(got - ref)) < 2e-4 * np.max(ref)


def test_window_is_torch_hamming():
    assert np.allclose(mfcc_np.hamming_periodic(400), torch.hamming_window(400).numpy().astype(np.float64), atol=1e-7)


def test_dct_is_orthonormal_and_mel_is_triangular():
    d = mfcc_np.dct_matrix()
    assert np.allclose(d.T @ d, np.eye(40), atol=1e-12)
    fb = mfcc_np.mel_filterbank()
    assert fb.shape == (257, 40) and fb.min() >= 0 and fb.max() <= 1.0
    peaks = fb.argmax(axis=0)
    assert np.all(np.diff(peaks) > 0)                        # HTK mel centres increase


def test_shapes_and_padding():
    c = mfcc_np.synth_clips(1, seed=3)[0]
    f = mfcc_np.mfcc_torchaudio(c)
    assert f.shape == (101, 40)
    assert mfcc_np.process_sound_feat(f).shape == (1, 100, 40)
    short = mfcc_np.mfcc_torchaudio(c[:8000])                # 51 frames -> zero-padded in MFCC domain
    p = mfcc_np.process_sound_feat(short)
    assert p.shape == (1, 100, 40) and np.all(p[0, 51:] == 0) and np.all(p[0, :51] == short)
    # a pure tone concentrates energy in one mel band
    t = np.arange(16000) / 16000.0
    tone = np.round(20000 * np.sin(2 * np.pi * 1000 * t)).astype(np.int16)
    mel = mfcc_np.power_spectrogram((tone / 32768.0).astype(np.float32)) @ mfcc_np.mel_filterbank()
    assert abs(int(mel[50].argmax()) - int(mfcc_np.mel_filterbank()[32].argmax())) <= 1
