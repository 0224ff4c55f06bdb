"""On-disk formats either side of the hot path: the reference's collected-triplet pickles
(pretext.py:82-92) and its wav ingest rules (Envs/audioLoader.py:101-145) -> the index-table pool the HIP step reads."""
import os
import pickle
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _write_wavs(tmp_path, task_num=3):
    from scipy.io import wavfile
    rng = np.random.default_rng(0)
    per_class = []
    for c in range(task_num):
        paths = []
        for j, n in enumerate([16000, 9000 + 500 * c, 20000, 12000][: 3 + (c % 2)]):     # one clip is too long (1.25 s)
            p = os.path.join(tmp_path, f"c{c}_{j}.wav")
            wavfile.write(p, 16000, rng.integers(-3000, 3000, size=n).astype(np.int16))
            paths.append(p)
        per_class.append(paths)
    return per_class


def test_wav_ingest_and_pickle_pool(tmp_path):
    import var_amd
    task_num = 3
    per_class = _write_wavs(str(tmp_path), task_num)
    pcm, lens, start, count = var_amd.load_wav_clips(per_class, max_sound_dur=1.0)
    assert pcm.dtype == np.int16 and pcm.shape[1] == 16000
    assert count.tolist() == [2, 3, 2] and start.tolist() == [0, 2, 5]          # the 1.25 s clips were skipped
    assert lens.tolist() == [16000, 9000, 16000, 9500, 12000, 16000, 10000]
    assert np.all(pcm[1, 9000:] == 0)                                            # zero padding beyond the clip
    capped = var_amd.load_wav_clips(per_class, max_sound_dur=1.0, load_size=1)
    assert capped[3].tolist() == [1, 1, 1]

    rng = np.random.default_rng(1)
    items = []
    for i in range(40):
        it = {"image": rng.integers(0, 256, size=(3, 84, 84), dtype=np.uint8),
              "ground_truth": np.array([rng.integers(0, task_num + 1)], dtype=np.int32)}
        if i % 2 == 0:
            it["sound_negative_id"] = np.array([rng.integers(0, task_num + 1)], dtype=np.int32)
        items.append(it)
    paths = []
    for k in range(2):
        p = os.path.join(str(tmp_path), f"data_{k}.pickle")
        with open(p, "wb") as f:
            pickle.dump(items[20 * k:20 * k + 20], f, protocol=pickle.HIGHEST_PROTOCOL)
        paths.append(p)
    pool = var_amd.TripletPool.from_pickles(paths, pcm, lens, start, count, task_num, seed=3, device="cpu")
    assert pool.n_items == 40 and pool.hw == 84 and pool.cpc is None             # ragged classes
    assert torch.equal(pool.images, torch.from_numpy(np.stack([it["image"] for it in items])))
    gt = np.array([int(it["ground_truth"][0]) for it in items])
    sn = pool.sn.numpy()
    assert np.array_equal(pool.gt.numpy(), gt)
    for i, it in enumerate(items):
        if "sound_negative_id" in it:
            assert sn[i] == int(it["sound_negative_id"][0])                      # stored id wins (dataset.py:70-72)
        else:
            assert sn[i] != gt[i] or sn[i] == task_num                            # collision -> "empty" (dataset.py:76-78)
            assert 0 <= sn[i] <= task_num
    pool.freeze_pairs()
    cls = torch.stack([pool.gt, pool.sn])
    ids = pool.clip_tab.long()
    for r in range(2):
        for i in range(40):
            c = int(cls[r, i])
            if c >= task_num:
                assert int(pool.len_tab[r, i]) == 0                              # "empty" class = zero-length clip
            else:
                assert start[c] <= int(ids[r, i]) < start[c] + count[c]          # a clip of the right class
                assert int(pool.len_tab[r, i]) == lens[int(ids[r, i])]
    tab = pool.index_table(8, 7, drop_last=False)
    assert tab.shape == (10, 40) and tab.dtype == torch.int32
    for e in range(2):                                                           # every epoch is a permutation
        assert sorted(tab[5 * e:5 * e + 5, :8].reshape(-1).tolist()) == list(range(40))
