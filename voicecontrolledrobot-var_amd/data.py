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


class SyntheticTripletPool:
    """HBM-resident pool of synthetic triplets: u8 images, int16 1 s clips per class, labels.

    Mirrors what VARDataset.__getitem__ (dataset.py:64-89) hands the loop: image, positive clip of
    class gt, negative clip of class sn (rule above), class taskNum = all-zero MFCC ("empty",
    dataset.py:37-38) which is encoded as clip length 0."""

    def __init__(self, n_items, hw=84, task_num=4, clips_per_class=64, seed=0, device="cuda", empty_frac=0.2):
        g = np.random.default_rng(seed)
        self.hw, self.task_num, self.device = hw, task_num, torch.device(device)
        self.images = torch.from_numpy(g.integers(0, 256, size=(n_items, 3, hw, hw), dtype=np.uint8)).to(self.device)
        gt = g.integers(0, task_num, size=n_items)
        gt[g.random(n_items) < empty_frac] = task_num          # 20 % "empty" positives
        sn = np.array([choose_negative_id(a, task_num, rand_int=lambda lo, hi: int(g.integers(lo, hi))) for a in gt])
        self.gt = torch.from_numpy(gt.astype(np.int64)).to(self.device)
        self.sn = torch.from_numpy(sn.astype(np.int64)).to(self.device)
        clips = synth_clips(task_num * clips_per_class, seed=seed + 1)
        self.clips = torch.from_numpy(clips).to(self.device)    # (task_num*cpc, 16000) int16, class-major
        self.cpc = clips_per_class
        self.n_items = n_items
        self._gen = torch.Generator(device=self.device)
        self._gen.manual_seed(seed + 2)

    # ---- fixed pairing + epoch permutation (what VARFineTuneDataset + DataLoader(shuffle=True) do,
    #      dataset.py:94-133,157-162), as index tables so that the step gathers straight from HBM ----
    def freeze_pairs(self):
        """Draw each item's positive / negative clip once (dataset.py:101-117) -> (2,N) int32 tables."""
        n = self.n_items
        cp = torch.randint(0, self.cpc, (2, n), device=self.device, generator=self._gen)
        cls = torch.stack([self.gt, self.sn])
        empty = cls >= self.task_num
        self.clip_tab = (torch.clamp(cls, max=self.task_num - 1) * self.cpc + cp).to(torch.int32).contiguous()
        self.len_tab = torch.where(empty, 0, self.clips.shape[1]).to(torch.int32).contiguous()
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

    def epoch_index_table(self, batch):
        """(steps, 5*batch) int32: rows [image_index | clip_index (2B) | lens (2B)] of one shuffled epoch
        (drop_last), built with a handful of device ops per EPOCH instead of per step."""
        steps = self.n_items // batch
        perm = torch.randperm(self.n_items, device=self.device, generator=self._gen)[:steps * batch]
        idx = perm.view(steps, batch)
        clip = self.clip_tab[:, perm].view(2, steps, batch).permute(1, 0, 2).reshape(steps, 2 * batch)
        lens = self.len_tab[:, perm].view(2, steps, batch).permute(1, 0, 2).reshape(steps, 2 * batch)
        return torch.cat([idx.to(torch.int32), clip, lens], dim=1).contiguous()

    def index_table(self, batch, min_rows):
        """At least `min_rows` step rows: whole shuffled epochs (epoch_index_table) back to back."""
        parts, rows = [], 0
        while rows < min_rows:
            parts.append(self.epoch_index_table(batch))
            rows += parts[-1].shape[0]
        return torch.cat(parts, dim=0).contiguous()

    def sample_indices(self, batch):
        """Random item ids and clip ids (device tensors, no host sync)."""
        idx = torch.randint(0, self.n_items, (batch,), device=self.device, generator=self._gen)
        cp = torch.randint(0, self.cpc, (2, batch), device=self.device, generator=self._gen)
        return idx, cp

    def gather(self, idx, cp, out_img=None, out_pcm=None, out_len=None):
        """image u8 (B,3,H,H), pcm int16 (2B,16000) [pos | neg], lens int32 (2B) (0 = empty class)."""
        gt, sn = self.gt[idx], self.sn[idx]
        img = torch.index_select(self.images, 0, idx, out=out_img)
        cls = torch.cat([gt, sn])
        empty = cls >= self.task_num
        clip_id = torch.clamp(cls, max=self.task_num - 1) * self.cpc + torch.cat([cp[0], cp[1]])
        pcm = torch.index_select(self.clips, 0, clip_id, out=out_pcm)
        lens = torch.where(empty, 0, self.clips.shape[1]).to(torch.int32)
        if out_len is not None:
            out_len.copy_(lens)
            lens = out_len
        return img, pcm, lens
