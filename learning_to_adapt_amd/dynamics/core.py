"""Shared host logic of the dynamics models (weights as PyTorch tensors, lazy HIP handle).

What the reference keeps in a TF graph + session (``dynamics/core/layers.py:115-171``,
``dynamics/core/utils.py:75-142``) is here a plain list of tensors per weight set in the
reference's parameter order - ``hidden_0/kernel [in,h0], hidden_0/bias, ..., output/kernel,
output/bias`` (``layers.py:160-163``) - plus a lazily created ``NativeModel`` that mirrors them
into HBM for the fused kernels.  Nothing here touches HIP before the first ``predict`` /
``get_actions`` (the reference forks env workers before it creates its session,
``samplers/sampler.py:37`` vs ``trainers/mb_trainer.py:46-48``).
"""

from collections import OrderedDict

import numpy as np
import torch

ACTIVATION_NAMES = (None, "identity", "relu", "tanh", "sigmoid", "swish")


def param_names(n_hidden):
    names = []
    for li in range(n_hidden):
        names += ["hidden_%d/kernel" % li, "hidden_%d/bias" % li]
    return names + ["output/kernel", "output/bias"]


def xavier_params(sizes, rng):
    """Xavier-uniform kernels, zero biases (``dynamics/core/utils.py:81-82``)."""
    params = []
    for fan_in, fan_out in zip(sizes[:-1], sizes[1:]):
        lim = np.sqrt(6.0 / (fan_in + fan_out))
        params.append(torch.from_numpy(rng.uniform(-lim, lim, size=(fan_in, fan_out)).astype(np.float32)))
        params.append(torch.zeros(fan_out, dtype=torch.float32))
    return params


def nonlinearity_name(fn):
    """The reference's models accept TensorFlow functions (``hidden_nonlinearity=tf.nn.tanh`` is the recorded default of
    ``RNNDynamicsModel``, ``rnn_dynamics.py:21``; ``run_rebal.py`` never overrides it) as well as strings.  A callable is
    resolved by its ``__name__`` - that is all a snapshot keeps of it (a pickled global) and all the drop-in needs."""
    if fn is None or isinstance(fn, str):
        return fn
    name = getattr(fn, "__name__", None)
    if callable(fn) and name in ("relu", "tanh", "sigmoid", "swish", "identity"):
        return name
    return fn       # left as it is: the constructors refuse it with their usual message


def torch_act(name):
    if name in (None, "identity"):
        return lambda x: x
    if name == "relu":
        return torch.relu
    if name == "tanh":
        return torch.tanh
    if name == "sigmoid":
        return torch.sigmoid
    if name == "swish":
        return lambda x: x * torch.sigmoid(x)
    raise ValueError("unsupported nonlinearity %r (supported: relu, tanh, sigmoid, swish, None)" % (name,))


def mlp_forward(x, params, hidden_act, output_act):
    """Stock-op forward used by ``fit`` / ``adapt`` (training is not the fused hot path)."""
    n_layers = len(params) // 2
    hid, out = torch_act(hidden_act), torch_act(output_act)
    for li in range(n_layers):
        x = x @ params[2 * li] + params[2 * li + 1]
        x = hid(x) if li < n_layers - 1 else out(x)
    return x


def as_param_list(params, n_hidden):
    """Accept the reference's ``OrderedDict`` (``layers.py:71-79``) or a flat list."""
    if isinstance(params, (dict, OrderedDict)):
        params = [params[k] for k in param_names(n_hidden)]
    out = []
    for p in params:
        t = p.detach().clone() if torch.is_tensor(p) else torch.from_numpy(np.array(p, dtype=np.float32))
        out.append(t.to(dtype=torch.float32, device="cpu").contiguous())
    assert len(out) == 2 * (n_hidden + 1)
    return out


def normalize(data, mean, std):
    return (data - mean) / (std + 1e-10)       # mlp_dynamics.py:265-266


def denormalize(data, mean, std):
    return data * (std + 1e-10) + mean         # mlp_dynamics.py:269-270


def training_device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


class TFAdam(torch.optim.Optimizer):
    """``tf.train.AdamOptimizer`` (the ``optimizer`` default of the reference's three dynamics models, e.g.
    ``mlp_dynamics.py:39,84-85``) as TensorFlow 1.13 documents its update:

        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t);  m = beta1 m + (1 - beta1) g;  v = beta2 v + (1 - beta2) g^2
        theta -= lr_t * m / (sqrt(v) + epsilon)

    ``epsilon`` is added to the UN-corrected root ("epsilon hat" of Kingma & Ba section 2), which is ``torch.optim.Adam``'s
    epsilon divided by ``sqrt(1 - beta2^t)`` - 32 x larger on the first step.  For weights whose gradient is below ~1e-6
    (dead units, inputs that do not matter yet) the two formulas take visibly different steps, which is what
    ``tests/test_fit_oracle.py`` measured before this class replaced ``torch.optim.Adam`` in the three ``fit`` loops."""

    def __init__(self, params, lr=1e-3, beta1=0.9, beta2=0.999, epsilon=1e-8):
        super().__init__(params, dict(lr=lr, beta1=beta1, beta2=beta2, epsilon=epsilon))

    @torch.no_grad()
    def step(self, closure=None):
        assert closure is None
        for group in self.param_groups:
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            gs = [p.grad for p in ps]
            for p in ps:
                st = self.state[p]
                if not st:
                    st["m"], st["v"] = torch.zeros_like(p), torch.zeros_like(p)
            ms = [self.state[p]["m"] for p in ps]
            vs = [self.state[p]["v"] for p in ps]
            t = group["t"] = group.get("t", 0) + 1
            b1, b2 = group["beta1"], group["beta2"]
            lr_t = group["lr"] * float(np.sqrt(1.0 - b2 ** t)) / (1.0 - b1 ** t)
            torch._foreach_mul_(ms, b1)
            torch._foreach_add_(ms, gs, alpha=1.0 - b1)
            torch._foreach_mul_(vs, b2)
            torch._foreach_addcmul_(vs, gs, gs, value=1.0 - b2)
            den = torch._foreach_sqrt(vs)
            torch._foreach_add_(den, group["epsilon"])
            torch._foreach_addcdiv_(ps, ms, den, value=-lr_t)


def norm_tensors(normalization, device):
    out = {}
    for key in ("obs", "act", "delta"):
        out[key] = (torch.as_tensor(np.asarray(normalization[key][0]), dtype=torch.float32, device=device),
                    torch.as_tensor(np.asarray(normalization[key][1]), dtype=torch.float32, device=device))
    return out
