"""MI355X-native VAR contrastive-pretext hot path (Kuka model).

Host-side mirror of the reference's plug-in seam: `config.pretextModel = VARPretextNet`
(Envs/pybullet/arms/tasks/fourInARow/config.py:30) and the body of
`VAR_Pretext.trainRepresentation` (VAR/pretext_VAR.py:16-95).  All arithmetic runs in
libvar_hip.so (hand-written gfx950 kernels) behind the C ABI of include/var_hip.h;
PyTorch supplies device memory, streams, autograd plumbing and torch.distributed (RCCL).
There is NO CPU fallback: without the built library or without a GPU the ops raise.
"""
from ._lib import VarHipError, build_library, library_path, load_library  # noqa: F401
from .layout import N_PARAMS, PARAM_SPECS  # noqa: F401
from .model import VARPretextNet  # noqa: F401
from .ithor import IthorTrainer, IthorVARPretextNet, project_representation  # noqa: F401
from .actor_critic import ArmNetPolicy  # noqa: F401
from .comm import RcclComm  # noqa: F401
from .trainer import VARTrainer, train_representation, train_representation_from_pool, multistep_lr  # noqa: F401
from .data import SyntheticTripletPool, TripletPool, choose_negative_id, load_wav_clips, process_sound_feat  # noqa: F401
from .ops import inbatch_contrastive_loss, mfcc, mfcc_psf, triplet_margin_loss  # noqa: F401
from .reward import IntrinsicReward, ReturnNormalizer, RunningMeanStd  # noqa: F401
