"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement, in numpy, of the audio front-end that the reference runs inside
its DataLoader workers:

  * Envs/audioLoader.py:147-157  get_mfcc(..., mfcc_from='torchaudio')
        torchaudio.transforms.MFCC(sample_rate=fs, n_mfcc=40, log_mels=True,
            melkwargs={n_fft:512, win_length:int(.025*fs)=400, hop_length:int(.01*fs)=160,
                       n_mels:40, f_min:0, f_max:None, window_fn:torch.hamming_window})
        int16 -> /32768. float32 (:154-155); output (40,T) transposed to (T,40) (:157)
  * Envs/audioLoader.py:241-252  processSoundFeat: leading axis, truncate to
        sound_dim[1] frames or zero-pad IN THE MFCC DOMAIN.

PARITY UNPINNED: the arithmetic lives in torchaudio (requirements.txt:11,
torchaudio~=0.12.1), which is absent from /root/reference and not installed in
the build container, and the reference holds no golden vectors for it.  The
restatement below follows torchaudio 0.12's published algorithm:

  Spectrogram : torch.stft(x, n_fft=512, hop=160, win_length=400,
                window=hamming_window(400, periodic=True) zero-padded (centred) to 512,
                center=True, pad_mode='reflect', normalized=False, onesided=True) -> |.|^2
  MelScale    : melscale_fbanks(n_freqs=257, f_min=0, f_max=sr/2, n_mels=40, norm=None,
                mel_scale='htk'); triangles = max(0, min(down_slope, up_slope))
  log         : log(mel + 1e-6)
  DCT-II ortho: create_dct(40, 40, 'ortho'): cos(pi/40*(n+.5)*k), row k=0 * 1/sqrt(2),
                all * sqrt(2/40); mfcc = mel^T @ dct
  frames      : T = 1 + N // 160  (101 for a 1 s clip), then truncate/pad to 100.

tests/test_oracle_mfcc.py pins the STFT/power stage of this file against
torch.stft (the very primitive torchaudio calls), which is the strongest pin
available here.
"""
import numpy as np

N_FFT = 512
WIN = 400
HOP = 160
N_MELS = 40
N_MFCC = 40
N_FREQ = N_FFT // 2 + 1
LOG_OFFSET = 1e-6


# per-dataset STFT parameters of Envs/audioLoader.py:23-31 at 16 kHz: (n_fft, win_length, hop_length)
DATASET_STFT = {"GoogleCommand": (512, 400, 160), "ESC50": (512, 400, 160), "FSC": (512, 400, 160),
                "Spatial": (512, 400, 160), "Synthetic": (512, 400, 160),
                "NSynth": (1024, 800, 640), "UrbanSound": (1024, 800, 640)}


def hamming_periodic(n=WIN, dtype=np.float64):
    # torch.hamming_window(n, periodic=True): 0.54 - 0.46 cos(2 pi i / n)
    i = np.arange(n, dtype=np.float64)
    return (0.54 - 0.46 * np.cos(2.0 * np.pi * i / n)).astype(dtype)


def padded_window(dtype=np.float64, n_fft=N_FFT, win=WIN):
    """The win-tap window centred in an n_fft-sample FFT frame (torch.stft semantics)."""
    w = np.zeros(n_fft, dtype=dtype)
    left = (n_fft - win) // 2
    w[left:left + win] = hamming_periodic(win, dtype)
    return w


def mel_filterbank(sample_rate=16000, dtype=np.float64, n_fft=N_FFT):
    """torchaudio.functional.melscale_fbanks(n_fft/2+1, 0, sr/2, 40, sr, norm=None, 'htk') -> (n_freq, 40)."""
    all_freqs = np.linspace(0.0, sample_rate // 2, n_fft // 2 + 1)
    m_min = 2595.0 * np.log10(1.0 + 0.0 / 700.0)
    m_max = 2595.0 * np.log10(1.0 + (sample_rate / 2.0) / 700.0)
    m_pts = np.linspace(m_min, m_max, N_MELS + 2)
    f_pts = 700.0 * (10.0 ** (m_pts / 2595.0) - 1.0)
    f_diff = f_pts[1:] - f_pts[:-1]
    slopes = f_pts[None, :] - all_freqs[:, None]            # (257, 42)
    down = (-1.0 * slopes[:, :-2]) / f_diff[:-1]
    up = slopes[:, 2:] / f_diff[1:]
    fb = np.maximum(0.0, np.minimum(down, up))
    return fb.astype(dtype)


def dct_matrix(dtype=np.float64):
    """torchaudio.functional.create_dct(40, 40, 'ortho') -> (n_mels, n_mfcc)."""
    n = np.arange(float(N_MELS))
    k = np.arange(float(N_MFCC))[:, None]
    dct = np.cos(np.pi / float(N_MELS) * (n + 0.5) * k)     # (n_mfcc, n_mels)
    dct[0] *= 1.0 / np.sqrt(2.0)
    dct *= np.sqrt(2.0 / float(N_MELS))
    return dct.T.astype(dtype)


def frames_reflect(x, n_fft=N_FFT, hop=HOP):
    """center=True / pad_mode='reflect' framing: (T, n_fft) windows of the padded signal."""
    n = x.shape[0]
    pad = n_fft // 2
    xp = np.pad(x, (pad, pad), mode="reflect")
    t = 1 + n // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(t)[:, None]
    return xp[idx]


def power_spectrogram(x, dtype=np.float64, n_fft=N_FFT, win=WIN, hop=HOP):
    """(T, n_fft/2+1) power spectrogram of a float waveform."""
    fr = frames_reflect(x.astype(dtype), n_fft, hop) * padded_window(dtype, n_fft, win)[None, :]
    spec = np.fft.rfft(fr.astype(np.float64), n=n_fft, axis=1)
    return (spec.real ** 2 + spec.imag ** 2).astype(dtype)


def mfcc_torchaudio(pcm, sample_rate=16000, dtype=np.float64, n_fft=N_FFT, win=WIN, hop=HOP):
    """int16 (or float) waveform -> (T, 40) MFCC, T = 1 + N//hop.

    Follows Envs/audioLoader.py:150-157 (int16 -> /32768 float32 first); (n_fft, win, hop) = the dataset's entry of
    audioLoader.param_dict (:23-31): 512/400/160 (GoogleCommand, FSC, ESC50 ...) or 1024/800/640 (NSynth, UrbanSound)."""
    pcm = np.asarray(pcm)
    if pcm.dtype == np.int16:
        x = (pcm / 32768.0).astype(np.float32)
    else:
        x = pcm.astype(np.float32)
    p = power_spectrogram(x, dtype, n_fft, win, hop)
    mel = p @ mel_filterbank(sample_rate, dtype, n_fft)
    logmel = np.log(mel + dtype(LOG_OFFSET))
    return (logmel @ dct_matrix(dtype)).astype(dtype)


def process_sound_feat(feat, sound_dim=(1, 100, 40)):
    """Envs/audioLoader.py:241-252: add axis, truncate or zero-pad to sound_dim[1] frames."""
    feat = np.expand_dims(feat, 0)
    nf = feat.shape[1]
    if sound_dim[1] < nf:
        return feat[:, :sound_dim[1], :]
    pad = np.zeros((sound_dim[0], sound_dim[1] - nf, sound_dim[2]), dtype=feat.dtype)
    return np.concatenate((feat, pad), axis=1)


def synth_clips(n, seed=0, n_samples=16000, lens=None):
    """Synthetic int16 clips of SURVEY section 8(d): 3000*N(0,1) + 8000*sin(2 pi f t), f~U(100,4000)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples) / 16000.0
    out = np.zeros((n, n_samples), dtype=np.int16)
    for i in range(n):
        f = rng.uniform(100.0, 4000.0)
        x = 3000.0 * rng.standard_normal(n_samples) + 8000.0 * np.sin(2 * np.pi * f * t)
        out[i] = np.round(np.clip(x, -32767, 32767)).astype(np.int16)
    return out


# ---- python_speech_features 0.6 (requirements.txt:13), the iTHOR / FSC branch of Envs/audioLoader.py:158-161 --------
# The package is absent from the build image, so this is a restatement of its published algorithm (base.py / sigproc.py
# of release 0.6) with the reference's call parameters: "parity unpinned" against the dependency itself.
def _round_half_up(x):
    return int(np.floor(x + 0.5))


def psf_filterbanks(nfilt=40, nfft=512, samplerate=16000, lowfreq=0, highfreq=None):
    highfreq = highfreq or samplerate / 2
    lowmel = 2595 * np.log10(1 + lowfreq / 700.)
    highmel = 2595 * np.log10(1 + highfreq / 700.)
    melpoints = np.linspace(lowmel, highmel, nfilt + 2)
    bins = np.floor((nfft + 1) * (700 * (10 ** (melpoints / 2595.0) - 1)) / samplerate)
    fb = np.zeros([nfilt, nfft // 2 + 1])
    for j in range(nfilt):
        for i in range(int(bins[j]), int(bins[j + 1])):
            fb[j, i] = (i - bins[j]) / (bins[j + 1] - bins[j])
        for i in range(int(bins[j + 1]), int(bins[j + 2])):
            fb[j, i] = (bins[j + 2] - i) / (bins[j + 2] - bins[j + 1])
    return fb


def mfcc_psf(signal, samplerate=16000, winlen=0.025, winstep=0.01, numcep=40, nfilt=40, nfft=512, preemph=0.97,
             ceplifter=22, append_energy=True):
    """mfcc(signal, fs, winlen=.025, winstep=.01, numcep=40, nfilt=40, nfft=512, winfunc=np.hamming) -> (T, 40) f64."""
    from scipy.fftpack import dct
    sig = np.asarray(signal)
    sig = np.append(sig[0], sig[1:] - preemph * sig[:-1])                       # sigproc.preemphasis
    frame_len, frame_step = _round_half_up(winlen * samplerate), _round_half_up(winstep * samplerate)
    slen = len(sig)
    numframes = 1 if slen <= frame_len else 1 + int(np.ceil((1.0 * slen - frame_len) / frame_step))
    padlen = int((numframes - 1) * frame_step + frame_len)
    padsignal = np.concatenate((sig, np.zeros((padlen - slen,))))
    idx = np.tile(np.arange(0, frame_len), (numframes, 1)) + \
        np.tile(np.arange(0, numframes * frame_step, frame_step), (frame_len, 1)).T
    frames = padsignal[idx.astype(np.int32)] * np.hamming(frame_len)            # sigproc.framesig
    pspec = 1.0 / nfft * np.square(np.absolute(np.fft.rfft(frames, nfft)))      # sigproc.powspec
    energy = np.sum(pspec, 1)
    energy = np.where(energy == 0, np.finfo(float).eps, energy)
    feat = np.dot(pspec, psf_filterbanks(nfilt, nfft, samplerate).T)
    feat = np.where(feat == 0, np.finfo(float).eps, feat)
    feat = np.log(feat)
    feat = dct(feat, type=2, axis=1, norm='ortho')[:, :numcep]
    n = np.arange(numcep)
    feat = (1 + (ceplifter / 2.) * np.sin(np.pi * n / ceplifter)) * feat        # lifter
    if append_energy:
        feat[:, 0] = np.log(energy)
    return feat
