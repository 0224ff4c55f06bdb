"""TEST INFRASTRUCTURE shared by the -m gpu suites: the torch restatement of one step and the ReLU-gate tracing that
turns "a unit within rounding of zero may flip" from a comment into an assertion."""
import numpy as np
import torch

from oracle import var_oracle as orc
from oracle.torch_oracle import KukaNetCPU


def relu_flips(ctx, net, image_f32, pos, neg, B):
    """Units whose ReLU gate differs between the HIP forward left in the workspace and the CPU network, with the CPU
    pre-activation of each: [(layer name, flat index, pre-activation)].  A gate can legitimately differ only where the
    pre-activation is within rounding of zero."""
    flips = []
    with torch.no_grad():
        x = image_f32
        for l in range(5):
            z = net.imgBranch[2 * l](x)
            got = ctx.debug_buffer(f"act{l + 1}").cpu()[:z.numel()].view(z.shape)
            bad = ((got > 0) != (z > 0)).nonzero(as_tuple=False)
            flips += [(f"imgBranch.{2 * l}", tuple(i.tolist()), float(z[tuple(i)])) for i in bad]
            x = torch.relu(z)
        z = net.imgTriplet[0](x.flatten(1))
        got = ctx.debug_buffer("hid_i").cpu()[:z.numel()].view(z.shape)
        flips += [("imgTriplet.0", tuple(i.tolist()), float(z[tuple(i)])) for i in ((got > 0) != (z > 0)).nonzero()]
        s = torch.cat([pos, neg])
        for l in range(4):
            z = net.soundCNN[2 * l](s)
            got = ctx.debug_buffer(f"sact{l + 1}").cpu()[:z.numel()].view(z.shape[0], 32, -1)
            bad = ((got > 0) != (z[..., 0] > 0)).nonzero(as_tuple=False)
            flips += [(f"soundCNN.{2 * l}", tuple(i.tolist()), float(z[..., 0][tuple(i)])) for i in bad]
            s = torch.relu(z)
        z = net.soundTriplet[0](s.flatten(1))
        got = ctx.debug_buffer("hid_s").cpu()[:z.numel()].view(z.shape)
        flips += [("soundTriplet.0", tuple(i.tolist()), float(z[tuple(i)])) for i in ((got > 0) != (z > 0)).nonzero()]
    return flips


def assert_grads_match_or_traced(var_amd, tr, net, g_ref, image_f32, pos, neg, B, tight=1e-3):
    """Every gradient tensor within `tight` (relative to the tensor's largest entry) -- or, if some tensor misses it, at
    least one ReLU gate differs between the two forwards, every differing gate has |pre-activation| < 1e-5, and the
    arena still agrees to 2e-2 in L2."""
    got = orc.unflatten_params(tr.grads.cpu().numpy())
    ref = orc.unflatten_params(g_ref)
    worst = {k: float(np.max(np.abs(got[k] - ref[k])) / (np.max(np.abs(ref[k])) + 1e-30)) for k, _ in orc.PARAM_SPECS}
    missed = {k: v for k, v in worst.items() if v >= tight}
    flips = relu_flips(tr.ctx, net, image_f32, pos, neg, B)
    for name, idx, z in flips:
        assert abs(z) < 1e-5, f"ReLU gate {name}{idx} differs at pre-activation {z}: not a rounding-level flip"
    if missed:
        assert flips, f"gradient tensors off by more than {tight} without any ReLU gate flip to explain it: {missed}"
        l2 = np.linalg.norm(tr.grads.cpu().numpy() - g_ref) / np.linalg.norm(g_ref)
        assert l2 < 2e-2, (l2, missed, flips)
    return worst, flips


def torch_loss_grad(sd, image_u8, pos, neg, hw=84):
    net = KukaNetCPU(hw)
    net.load_state_dict({k: torch.as_tensor(np.asarray(v)) for k, v in sd.items()})
    image = (image_u8 / 255.).float()
    a, p, n = net(image, pos, neg)
    loss = torch.nn.TripletMarginLoss(margin=1.0, p=2)(a, p, n)
    loss.backward()
    g = torch.cat([dict(net.named_parameters())[k].grad.reshape(-1) for k, _ in orc.PARAM_SPECS]).numpy()
    return net, float(loss.detach()), g, (a.detach().numpy(), p.detach().numpy(), n.detach().numpy()), image


