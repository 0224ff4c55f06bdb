"""Frozen-encoder inference for the RL stage: embeddings, intrinsic reward and return-normalised reward
(BASELINE config 5).

Mirrors what the reference's vectorised-env wrapper does around the pretext model, without the simulator
plumbing: Envs/vec_env/vec_pretext_normalize.py:82-101 (getEmbeddings / calcReward), :47-59 (discounted-return
normalisation with clipping) and Envs/vec_env/running_mean_std.py:4-35 (streaming mean / variance).  The encoder
forward is the HIP path (VARPretextNet.forward); the rest is host arithmetic on (num_envs,) arrays in float64,
as in the reference.
"""
import numpy as np
import torch


class RunningMeanStd:
    """Streaming mean / variance of batches (parallel-variance update), running_mean_std.py:4-35."""

    def __init__(self, epsilon=1e-4, shape=()):
        self.mean = np.zeros(shape, np.float64)
        self.var = np.ones(shape, np.float64)
        self.count = epsilon

    def update(self, arr):
        arr = np.asarray(arr)
        self.update_from_moments(arr.mean(axis=0), arr.var(axis=0), arr.shape[0])

    def update_from_moments(self, batch_mean, batch_var, batch_count):
        delta = batch_mean - self.mean
        total = self.count + batch_count
        m2 = self.var * self.count + batch_var * batch_count + np.square(delta) * self.count * batch_count / total
        self.mean = self.mean + delta * batch_count / total
        self.var = m2 / total
        self.count = total


class ReturnNormalizer:
    """rew / sqrt(var(discounted return) + eps), clipped; returns reset where an episode ended
    (vec_pretext_normalize.py:47-59: gamma 0.99, cliprew 10, epsilon 1e-8)."""

    def __init__(self, num_envs, gamma=0.99, cliprew=10.0, epsilon=1e-8):
        self.ret_rms = RunningMeanStd(shape=())
        self.ret = np.zeros(num_envs)
        self.gamma, self.cliprew, self.epsilon = gamma, cliprew, epsilon

    def __call__(self, rews, news):
        rews = np.asarray(rews, dtype=np.float64)
        self.ret = self.ret * self.gamma + rews
        self.ret_rms.update(self.ret)
        out = np.clip(rews / np.sqrt(self.ret_rms.var + self.epsilon), -self.cliprew, self.cliprew)
        self.ret[np.asarray(news, dtype=bool)] = 0.0
        return out

    def reset(self):
        self.ret[:] = 0.0


class IntrinsicReward:
    """reward = env_reward + <image_feat, goal_sound_feat> (+ <current_sound_feat, goal_sound_feat> when
    sound_sound is on): getEmbeddings + calcReward of the reference wrapper, on the frozen HIP encoder.

    The goal sound of an episode is embedded once: pass goal_sound=None afterwards and the encoder returns the cached
    embedding (the reference signals this with an all-inf tensor, pretext_base.py:29-32; both are accepted)."""

    def __init__(self, model, representation_dim=3, sound_sound=False):
        self.model = model.eval()
        self.rep = representation_dim
        self.sound_sound = sound_sound

    @torch.no_grad()
    def embeddings(self, image_u8, goal_sound=None, current_sound=None):
        dev = next(self.model.parameters()).device
        to = lambda a: None if a is None else torch.as_tensor(np.ascontiguousarray(a)).to(dev)  # noqa: E731
        img = to(image_u8)
        if img.dtype != torch.uint8:                  # the wrapper divides by 255 on the host; u8 goes in as is
            img = img.float().contiguous()
        goal = to(goal_sound)
        if goal is None:
            b = img.shape[0]
            goal = torch.full((b, 1, 100, 40), float("inf"), device=dev)
        d = self.model(img, goal.float().contiguous(),
                       to(current_sound).float().contiguous() if (self.sound_sound and current_sound is not None) else None)
        image_feat = d["image_feat"].cpu().numpy()
        goal_feat = d["sound_feat_positive"].cpu().numpy()
        cur_feat = d["sound_feat_negative"].cpu().numpy() if (self.sound_sound and current_sound is not None) else 0.0
        return image_feat, goal_feat, cur_feat

    # ---- latency path: the whole frozen-encoder step as a replayed HIP graph --------------------------------
    def capture(self, batch):
        """Capture the frozen encoder for a fixed number of envs (BASELINE config 5: 8) into two HIP graphs --
        image + goal sound (first step of an episode) and image only (later steps: the goal embedding is reused) --
        over static device buffers.  Weights are packed once here: call again after loading another checkpoint.
        Returns self; then use step()."""
        from ._lib import Context, current_stream_handle, new_graph, ptr
        m = self.model
        flat = m.flat_parameters()
        dev = flat.device
        c = Context.get(dev.index)
        hw = m.config.img_dim[1]
        B = int(batch)
        c.ensure_plan(B, hw)
        self._B = B
        self._img = torch.zeros((B, 3, hw, hw), dtype=torch.uint8, device=dev)
        self._goal = torch.zeros((B, 1, 100, 40), dtype=torch.float32, device=dev)
        self._image_feat = torch.zeros((B, 3), device=dev)
        self._goal_feat = torch.zeros((B, 3), device=dev)
        self._reward = torch.zeros((B,), device=dev)
        w = m.hip_weights(c, force=True)        # the frozen model's own packed image; the graphs keep its address

        def body(with_goal):
            w.bind()
            if not with_goal and B <= 32:
                # the goal embedding is cached: <image_feat, goal_feat> rides in the image head's launch (var_set_reward_dot)
                c.check(c.lib.var_set_reward_dot(c.handle, ptr(self._goal_feat), ptr(self._reward)), "var_set_reward_dot")
            c.check(c.lib.var_arm_encoder_fwd(c.handle, current_stream_handle(), ptr(flat), ptr(self._img), 1,
                                              self._img.stride(0), ptr(self._goal) if with_goal else None, None, B, hw,
                                              ptr(self._image_feat), ptr(self._goal_feat) if with_goal else None,
                                              None, None, None, 2), "var_arm_encoder_fwd")      # 2: inference, small-batch kernels
            if not with_goal and B <= 32:
                return
            # <image_feat, goal_feat> (calcReward's torch.sum(a * b, dim=1): one launch instead of a product and a reduction)
            c.check(c.lib.var_row_dot(c.handle, current_stream_handle(), ptr(self._image_feat), ptr(self._goal_feat), B,
                                      self._image_feat.shape[1], ptr(self._reward)), "var_row_dot")

        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        self._graphs = {}
        with torch.cuda.stream(side):
            for with_goal in (True, False):
                body(with_goal)                                  # warm-up outside capture (lazy kernel attributes)
                g = new_graph()
                with torch.cuda.graph(g, stream=side, capture_error_mode="thread_local"):
                    body(with_goal)
                self._graphs[with_goal] = g
        torch.cuda.current_stream().wait_stream(side)
        return self

    def step(self, image_u8, goal_sound=None):
        """One env step of `batch` envs: copies the observations into the static buffers and replays the graph.
        Returns device tensors (image_feat (B,3), goal_feat (B,3), <image_feat, goal_feat> (B,)); they are
        overwritten by the next step."""
        self._img.copy_(torch.as_tensor(image_u8), non_blocking=True)
        if goal_sound is not None:
            self._goal.copy_(torch.as_tensor(goal_sound), non_blocking=True)
        self._graphs[goal_sound is not None].replay()
        return self._image_feat, self._goal_feat, self._reward

    def reward(self, env_reward, image_feat, goal_feat, cur_feat=0.0):
        img_sound = np.sum(image_feat[:, :self.rep] * goal_feat, axis=1)
        snd_sound = np.sum(cur_feat * goal_feat, axis=1) if self.sound_sound else np.zeros_like(img_sound)
        return img_sound + snd_sound * float(self.sound_sound) + np.asarray(env_reward), img_sound, snd_sound
