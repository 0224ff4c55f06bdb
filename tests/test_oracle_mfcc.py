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


def test_psf_mfcc_oracle_known_properties():
    """python_speech_features branch (Envs/audioLoader.py:158-161): frame count, energy coefficient, silence floor,
    filterbank shape -- properties of the published algorithm (the package itself is absent: parity unpinned)."""
    from oracle import mfcc_np
    rng = np.random.default_rng(3)
    x = np.round(3000 * rng.standard_normal(16000)).astype(np.int16)
    f = mfcc_np.mfcc_psf(x)
    assert f.shape == (1 + int(np.ceil((16000 - 400) / 160)), 40) and f.dtype == np.float64
    assert mfcc_np.mfcc_psf(x[:300]).shape == (1, 40)
    fb = mfcc_np.psf_filterbanks()
    assert fb.shape == (40, 257) and fb.min() >= 0 and fb.max() <= 1.0
    # coefficient 0 is the log of the frame energy = log(sum |rfft|^2 / 512) of the pre-emphasised, windowed frame
    y = np.append(x[0], x[1:] - 0.97 * x[:-1].astype(np.float64))
    fr = y[160:560] * np.hamming(400)
    e = np.sum(np.abs(np.fft.rfft(fr, 512)) ** 2) / 512
    assert abs(f[1, 0] - np.log(e)) < 1e-9
    # silence: every filterbank energy is replaced by eps before the log
    z = mfcc_np.mfcc_psf(np.zeros(1000, dtype=np.int16))
    assert np.allclose(z[:, 0], np.log(np.finfo(float).eps))
    # scaling the signal by c adds 2 log c to coefficient 0 and sqrt(40)*2 log c to ... only the DC cepstrum: c1.. unchanged
    f2 = mfcc_np.mfcc_psf((x // 2 * 2).astype(np.int16))
    f4 = mfcc_np.mfcc_psf(((x // 2 * 2) // 2).astype(np.int16))
    assert np.allclose(f2[:, 1:], f4[:, 1:], atol=1e-9) and np.allclose(f2[:, 0] - f4[:, 0], 2 * np.log(2.0), atol=1e-9)
