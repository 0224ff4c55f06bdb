"""Batch assembly rules of the reference's data layer (dataset.py, Envs/audioLoader.py) and the
device-resident synthetic triplet pool used by bench.py and the tests.  Host logic only."""
import numpy as np
import torch


def choose_negative_id(gt, task_num, stored=None, rand_int=None):
    """dataset.py:70-78: use the stored `sound_negative_id` if present, else draw
    randint(0, taskNum) and map a collision with the label to taskNum (the "empty" class)."""
    if stored is not None:
        return int(stored)
    if rand_int is None:
        rand_int = lambda lo, hi: int(torch.randint(low=lo, high=hi, size=()).item())
    sn = rand_int(0, task_num)
    return task_num if sn == int(gt) else sn


def process_sound_feat(feat, sound_dim=(1, 100, 40)):
    """Envs/audioLoader.py:241-252: add a leading axis, truncate to sound_dim[1] frames or
    zero-pad in the MFCC domain."""
    feat = np.expand_dims(np.asarray(feat), 0)
    nf = feat.shape[1]
    if sound_dim[1] < nf:
        return feat[:, :sound_dim[1], :]
    pad = np.zeros((sound_dim[0], sound_dim[1] - nf, sound_dim[2]), dtype=feat.dtype)
    return np.concatenate((feat, pad), axis=1)


def synth_clips(n, seed=0, n_samples=16000):
    """SURVEY section 8(d) synthetic audio: int16 round(clip(3000*N(0,1) + 8000*sin(2 pi f t))), f~U(100,4000)."""
    rng = np.random.default_rng(seed)
    t = np.arange(n_samples) / 16000.0
    out = np.zeros((n, n_samples), dtype=np.int16)
    for i in range(n):
        f = rng.uniform(100.0, 4000.0)
        x = 3000.0 * rng.standard_normal(n_samples) + 8000.0 * np.sin(2 * np.pi * f * t)
        out[i] = np.round(np.clip(x, -32767, 32767)).astype(np.int16)
    return out


def load_wav_clips(paths_per_class, max_sound_dur=1.0, load_size=None, n_samples=16000):
    """Wav ingest of Envs/audioLoader.py:101-145 (load2Words): per class, read the files in order with
    scipy.io.wavfile, skip clips longer than max_sound_dur seconds, stop after load_size clips.  Returns
    (pcm int16 (M, n_samples) zero-padded, lens int32 (M), class_start, class_count) with the clips class-major."""
    from scipy.io import wavfile
    pcm, lens, start, count = [], [], [], []
    for paths in paths_per_class:
        start.append(len(pcm))
        for path in paths:
            fs, x = wavfile.read(path)
            if x.size / fs > max_sound_dur:
                continue
            if x.dtype != np.int16 or x.ndim != 1:
                raise ValueError(f"{path}: expected mono int16 PCM (audioLoader.py:154 divides by 32768), got {x.dtype} {x.shape}")
            row = np.zeros(n_samples, dtype=np.int16)
            row[:x.size] = x[:n_samples]
            pcm.append(row)
            lens.append(min(x.size, n_samples))
            if load_size is not None and len(pcm) - start[-1] >= load_size:
                break
        count.append(len(pcm) - start[-1])
    return np.stack(pcm), np.asarray(lens, dtype=np.int32), np.asarray(start), np.asarray(count)


class TripletPool:
    """HBM-resident pool of triplets: u8 images, int16 clips per class, labels.

    Mirrors what VARDataset.__getitem__ (dataset.py:64-89) hands the loop: image, positive clip of class gt,
    negative clip of class sn (rule above), class taskNum = all-zero MFCC ("empty", dataset.py:37-38) which is
    encoded as clip length 0.  Built from arrays (this constructor), from the reference's collected pickles
    (from_pickles) or synthetically (SyntheticTripletPool)."""

    def __init__(self, images, gt, sn, clips, clip_len, class_start, class_count, task_num, seed=0, device="cuda"):
        self.device = torch.device(device)
        as_t = lambda a, dt: torch.as_tensor(np.ascontiguousarray(a) if isinstance(a, np.ndarray) else a).to(dt).to(self.device)  # noqa: E731
        self.images = as_t(images, torch.uint8)                 # (N,3,H,H)
        self.hw, self.task_num = int(self.images.shape[2]), int(task_num)
        self.gt, self.sn = as_t(gt, torch.int64), as_t(sn, torch.int64)
        self.clips = as_t(clips, torch.int16)                   # (M, n) class-major
        self.clip_len = as_t(clip_len, torch.int32)             # valid samples per clip
        self.class_start, self.class_count = as_t(class_start, torch.int64), as_t(class_count, torch.int64)
        if int(self.class_count.min()) < 1:
            raise ValueError("every class needs at least one clip")
        cnt = self.class_count.tolist()
        self.cpc = cnt[0] if all(c == cnt[0] for c in cnt) and self.class_start.tolist() == [i * cnt[0] for i in range(len(cnt))] else None
        self.n_items = int(self.images.shape[0])
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed + 2)

    @classmethod
    def from_pickles(cls, paths, clips, clip_len, class_start, class_count, task_num, seed=0, device="cuda",
                     load_num=None):
        """The reference's collected data: each pickle is a list of dicts {'image': u8 (3,H,W), 'ground_truth':
        int (1,), 'sound_negative_id': int (1,) (optional)} (pretext.py:82-92).  A missing negative id is drawn
        by the rule of dataset.py:73-78 (choose_negative_id).  load_num: as loadEnvData's `loadNum`
        (dataset.py:150-154) -- a random subset of that many pickle files instead of all of them ('all' / None)."""
        import pickle
        paths = list(paths)
        if load_num is not None and load_num != 'all' and int(load_num) < len(paths):
            pick = np.random.default_rng(seed + 7).choice(len(paths), size=int(load_num), replace=False)
            paths = [paths[i] for i in sorted(pick.tolist())]
        items = []
        for path in paths:
            with open(path, "rb") as f:                         # the authors' own format; only load trusted files
                items.extend(pickle.load(f))
        rng = np.random.default_rng(seed)
        images = np.stack([np.asarray(it["image"], dtype=np.uint8) for it in items])
        gt = np.array([int(np.asarray(it["ground_truth"]).reshape(-1)[0]) for it in items])
        sn = np.array([choose_negative_id(g, task_num,
                                          stored=None if it.get("sound_negative_id") is None
                                          else int(np.asarray(it["sound_negative_id"]).reshape(-1)[0]),
                                          rand_int=lambda lo, hi: int(rng.integers(lo, hi)))
                       for g, it in zip(gt, items)])
        return cls(images, gt, sn, clips, clip_len, class_start, class_count, task_num, seed=seed, device=device)

    def set_datasets(self, ds_start, ds_count):
        """Two-level clip draw of Envs/audioLoader.py:174-176: the clips of a class come from several datasets
        (GoogleCommand, NSynth, ...); the reference first picks a DATASET uniformly, then a clip inside it, so clips
        of a small dataset are drawn more often than a flat draw over the class would.  ds_start / ds_count:
        (n_classes, n_datasets) first clip row and clip count of each (class, dataset) block of `clips` (count 0 = the
        dataset has no clip of that class and is never picked)."""
        self.ds_start = torch.as_tensor(np.asarray(ds_start)).to(torch.int64).to(self.device)
        self.ds_count = torch.as_tensor(np.asarray(ds_count)).to(torch.int64).to(self.device)
        if self.ds_start.shape != self.ds_count.shape or int((self.ds_count > 0).sum(1).min()) < 1:
            raise ValueError("every class needs a clip in at least one dataset")
        return self

    def _clip_ids(self, cls_ids, shape):
        """A random clip of each class id (class taskNum -> any clip, its length is forced to 0)."""
        c = torch.clamp(cls_ids, max=self.task_num - 1)
        if getattr(self, "ds_count", None) is not None:         # dataset first, then clip (audioLoader.py:174-176)
            cnt = self.ds_count[c]                               # (..., n_datasets)
            avail = (cnt > 0).to(torch.float32)
            u = torch.rand(shape, device=self.device, generator=self._gen)
            k = torch.minimum((u * avail.sum(-1)).long(), avail.sum(-1).long() - 1)      # k-th non-empty dataset
            ds = ((torch.cumsum(avail, -1) - 1 == k[..., None]) & (cnt > 0)).to(torch.int64).argmax(-1)
            n = torch.gather(cnt, -1, ds[..., None])[..., 0]
            v = torch.rand(shape, device=self.device, generator=self._gen)
            return torch.gather(self.ds_start[c], -1, ds[..., None])[..., 0] + torch.minimum((v * n).long(), n - 1)
        if self.cpc is not None:                                # uniform classes: the draw the synthetic pool always made
            cp = torch.randint(0, self.cpc, shape, device=self.device, generator=self._gen)
        else:
            u = torch.rand(shape, device=self.device, generator=self._gen)
            cp = torch.minimum((u * self.class_count[c]).long(), self.class_count[c] - 1)
        return self.class_start[c] + cp


    # ---- fixed pairing + epoch permutation (what VARFineTuneDataset + DataLoader(shuffle=True) do,
    #      dataset.py:94-133,157-162), as index tables so that the step gathers straight from HBM ----
    def freeze_pairs(self):
        """Draw each item's positive / negative clip once (dataset.py:101-117) -> (2,N) int32 tables."""
        n = self.n_items
        cls = torch.stack([self.gt, self.sn])
        empty = cls >= self.task_num
        ids = self._clip_ids(cls, (2, n))
        self.clip_tab = ids.to(torch.int32).contiguous()
        self.len_tab = torch.where(empty, 0, self.clip_len[ids]).to(torch.int32).contiguous()
        self._perm = None
        self._cursor = 0
        return self

    def next_batch_indices(self, batch):
        """(image_index (B), clip_index (2B), lens (2B)) int32 for the next shuffled batch; 2 small gathers."""
        if self._perm is None or self._cursor + batch > self.n_items:
            self._perm = torch.randperm(self.n_items, device=self.device, generator=self._gen)
            self._perm32 = self._perm.to(torch.int32)
            self._cursor = 0
        sl = slice(self._cursor, self._cursor + batch)
        self._cursor += batch
        idx = self._perm[sl]
        return self._perm32[sl], self.clip_tab[:, idx].reshape(-1), self.len_tab[:, idx].reshape(-1)

    def steps_per_epoch(self, batch, drop_last=False):
        """len(DataLoader): ceil(N / batch) with the reference's drop_last=False (VAR/pretext_VAR.py:24)."""
        return self.n_items // batch if drop_last else -(-self.n_items // batch)

    def tail_batch(self, batch, drop_last=False):
        """Size of the short last batch of an epoch (0 = none): 300 triplets at batch 128 -> 44."""
        return 0 if drop_last else self.n_items % batch

    def epoch_index_table(self, batch, drop_last=False):
        """(steps, 5*batch) int32: rows [image_index | clip_index (2B) | lens (2B)] of one shuffled epoch, built with
        a handful of device ops per EPOCH instead of per step.  drop_last=False (the reference's DataLoader,
        dataset.py:157-162): when batch does not divide the pool the last row is the SHORT batch -- its Bt =
        tail_batch(batch) samples packed at the head of the row in the same layout [image_index (Bt) | clip_index
        (2 Bt) | lens (2 Bt)], the rest of the row zero (VARTrainer.capture_epoch_steps(tail_batch=Bt) runs it)."""
        full = self.n_items // batch
        perm = torch.randperm(self.n_items, device=self.device, generator=self._gen)

        def rows(p, steps, b):
            idx = p.view(steps, b)
            clip = self.clip_tab[:, p].view(2, steps, b).permute(1, 0, 2).reshape(steps, 2 * b)
            lens = self.len_tab[:, p].view(2, steps, b).permute(1, 0, 2).reshape(steps, 2 * b)
            return torch.cat([idx.to(torch.int32), clip, lens], dim=1)

        tab = rows(perm[:full * batch], full, batch)
        bt = self.tail_batch(batch, drop_last)
        if bt:
            tail = torch.zeros((1, 5 * batch), dtype=torch.int32, device=self.device)
            tail[:, :5 * bt] = rows(perm[full * batch:], 1, bt)
            tab = torch.cat([tab, tail], dim=0)
        return tab.contiguous()

    def index_table(self, batch, min_rows, drop_last=True):
        """At least `min_rows` step rows: whole shuffled epochs (epoch_index_table) back to back.  Full rows only by
        default: a table with short last rows (drop_last=False) must be handed to capture_epoch_steps together with
        tail_batch / steps_per_epoch -- run as full batches its zero-padded tail rows would train on (image 0, clip 0,
        length 0) and silently skew loss and throughput."""
        parts, rows = [], 0
        while rows < min_rows:
            parts.append(self.epoch_index_table(batch, drop_last))
            rows += parts[-1].shape[0]
        return torch.cat(parts, dim=0).contiguous()

    def sample_indices(self, batch):
        """Random item ids and clip choices (device tensors, no host sync)."""
        idx = torch.randint(0, self.n_items, (batch,), device=self.device, generator=self._gen)
        if self.cpc is not None:
            cp = torch.randint(0, self.cpc, (2, batch), device=self.device, generator=self._gen)
        else:
            cp = torch.rand((2, batch), device=self.device, generator=self._gen)
        return idx, cp

    def gather(self, idx, cp, out_img=None, out_pcm=None, out_len=None):
        """image u8 (B,3,H,H), pcm int16 (2B,n) [pos | neg], lens int32 (2B) (0 = empty class)."""
        gt, sn = self.gt[idx], self.sn[idx]
        img = torch.index_select(self.images, 0, idx, out=out_img)
        cls = torch.cat([gt, sn])
        empty = cls >= self.task_num
        c = torch.clamp(cls, max=self.task_num - 1)
        cpf = torch.cat([cp[0], cp[1]])
        if self.cpc is not None:
            clip_id = c * self.cpc + cpf
        else:
            clip_id = self.class_start[c] + torch.minimum((cpf * self.class_count[c]).long(), self.class_count[c] - 1)
        pcm = torch.index_select(self.clips, 0, clip_id, out=out_pcm)
        lens = torch.where(empty, 0, self.clip_len[clip_id]).to(torch.int32)
        if out_len is not None:
            out_len.copy_(lens)
            lens = out_len
        return img, pcm, lens


class SyntheticTripletPool(TripletPool):
    """Synthetic pool: U{0..255} images, sine+noise clips, 20 % "empty" positives.  n_samples: clip length
    (16 000 = 1 s, the Kuka / GoogleCommand shape; up to 96 000 = the iTHOR / FluentSpeech 6 s clips,
    Envs/ai2thor/config.py:119); ragged_lens: draw each clip's valid length in [n_samples / 2, n_samples]."""

    def __init__(self, n_items, hw=84, task_num=4, clips_per_class=64, seed=0, device="cuda", empty_frac=0.2,
                 n_samples=16000, ragged_lens=False):
        g = np.random.default_rng(seed)
        images = g.integers(0, 256, size=(n_items, 3, hw, hw), dtype=np.uint8)
        gt = g.integers(0, task_num, size=n_items)
        gt[g.random(n_items) < empty_frac] = task_num          # 20 % "empty" positives
        sn = np.array([choose_negative_id(a, task_num, rand_int=lambda lo, hi: int(g.integers(lo, hi))) for a in gt])
        clips = synth_clips(task_num * clips_per_class, seed=seed + 1, n_samples=n_samples)   # (task_num*cpc, n) int16, class-major
        lens = np.full(clips.shape[0], clips.shape[1], dtype=np.int32)
        if ragged_lens:
            lens = g.integers(n_samples // 2, n_samples + 1, size=clips.shape[0]).astype(np.int32)
            for i, n in enumerate(lens):
                clips[i, n:] = 0
        super().__init__(images, gt, sn, clips, lens,
                         np.arange(task_num) * clips_per_class, np.full(task_num, clips_per_class), task_num,
                         seed=seed, device=device)
