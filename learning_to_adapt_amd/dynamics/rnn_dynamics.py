"""``RNNDynamicsModel`` - drop-in for ``learning_to_adapt/dynamics/rnn_dynamics.py:11-353`` (ReBAL).

Same constructor keywords as the reference (``run_scripts/run_rebal.py:23-32``): ``name, env,
hidden_sizes=(512,), cell_type='lstm', hidden_nonlinearity, output_nonlinearity, batch_size,
learning_rate, normalize_input, optimizer, valid_split_ratio, rolling_average_persitency,
backprop_steps``; same ``predict(obs, act, hidden) -> (pred_obs, next_hidden)``,
``get_initial_hidden(batch_size)``, ``fit`` on ``[paths, T, dim]`` arrays, ``normalization`` and
pickling surface.  Nonlinearities are given by name (``'tanh'`` = the reference default
``tf.nn.tanh``).

Cells: everything ``create_rnn`` (``dynamics/core/utils.py:192-236``) builds - ``cell_type`` ``'lstm'``
(the run script's), ``'gru'``, ``'rnn'`` and stacks (several entries in ``hidden_sizes`` = ``MultiRNNCell``);
see ``dynamics/rnn_cells.py``.  The hidden state has the reference's structure (``:273-293``): an
``LSTMStateTuple(c, h)`` of float32 arrays ``[batch, units]`` for one LSTM layer, a plain array for one GRU / RNN
layer, a list / tuple of those for a stack.  One LSTM layer of 128 / 256 / 512 units runs on the fused MFMA kernel,
every other configuration on the generic VALU kernel (``csrc/l2a_rnn_valu.h``).

* ``predict`` runs on the MI355X through ``l2a_lstm_predict``.
* ``RNNMPCController`` asks for the ``NativeLSTM`` handle (``planner_model()``) and launches the
  fused recurrent rollout; it never calls ``predict`` per horizon step.
* ``fit`` is stock PyTorch-ROCm: truncated BPTT over ``backprop_steps`` chunks, gradients averaged
  over the chunks of a batch and applied once with Adam (reference ``:165-190``), the same
  rolling-average early stop.  Training is not part of the fused hot path.
"""

import time
from collections import OrderedDict, namedtuple

import numpy as np
import torch

from ..utils.serializable import Serializable
from . import core, rnn_cells
from .rnn_cells import LSTMStateTuple, FORGET_BIAS  # noqa: F401  (re-exported)

PARAM_NAMES = ("rnn/lstm_cell/kernel", "rnn/lstm_cell/bias", "output/kernel", "output/bias")   # one LSTM layer


def glorot_lstm_params(obs_dim, act_dim, units, rng):
    """TF defaults: glorot-uniform kernels, zero biases (``dynamics/core/utils.py:155-156,197``)."""
    k_in = obs_dim + act_dim + units
    lim = np.sqrt(6.0 / (k_in + 4 * units))
    kernel = torch.from_numpy(rng.uniform(-lim, lim, size=(k_in, 4 * units)).astype(np.float32))
    lim = np.sqrt(6.0 / (units + obs_dim))
    wout = torch.from_numpy(rng.uniform(-lim, lim, size=(units, obs_dim)).astype(np.float32))
    return [kernel, torch.zeros(4 * units), wout, torch.zeros(obs_dim)]


def lstm_forward(x_seq, c, h, params, cell_act, output_act):
    """Stock-op LSTM over ``x_seq [B, T, in]`` (used by ``fit``).  Returns ``(deltas [B, T, obs], c, h)``."""
    act, out = core.torch_act(cell_act), core.torch_act(output_act)
    kernel, bias, wout, bout = params
    ys = []
    for t in range(x_seq.shape[1]):
        z = torch.cat([x_seq[:, t], h], dim=1) @ kernel + bias
        i, j, f, o = torch.chunk(z, 4, dim=1)
        c = torch.sigmoid(f + FORGET_BIAS) * c + torch.sigmoid(i) * act(j)
        h = torch.sigmoid(o) * act(c)
        ys.append(out(h @ wout + bout))
    return torch.stack(ys, dim=1), c, h


class RNNDynamicsModel(Serializable):
    """Recurrent model of normalised state deltas."""

    _activations = core.ACTIVATION_NAMES

    def __init__(self,
                 name,
                 env,
                 hidden_sizes=(512,),
                 cell_type="lstm",
                 hidden_nonlinearity="tanh",
                 output_nonlinearity=None,
                 batch_size=500,
                 learning_rate=0.001,
                 normalize_input=True,
                 optimizer=None,
                 valid_split_ratio=0.2,
                 rolling_average_persitency=0.99,
                 backprop_steps=50,
                 init_seed=None,
                 ):
        Serializable.quick_init(self, locals())
        self.recurrent = True

        self.normalization = None
        self.normalize_input = normalize_input
        self.valid_split_ratio = valid_split_ratio
        self.rolling_average_persitency = rolling_average_persitency
        self.backprop_steps = backprop_steps
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.name = name
        self._dataset_train = None
        self._dataset_test = None

        self.obs_space_dims = int(env.observation_space.shape[0])
        self.action_space_dims = int(env.action_space.shape[0])

        if cell_type not in rnn_cells.CELL_TYPES:
            raise NotImplementedError("cell_type %r (reference dynamics/core/utils.py:199-213 knows %s)"
                                      % (cell_type, rnn_cells.CELL_TYPES))
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        if not 1 <= len(self.hidden_sizes) <= 4:
            raise NotImplementedError("1 to 4 stacked cells are supported (hidden_sizes=%r)" % (hidden_sizes,))
        hidden_nonlinearity = core.nonlinearity_name(hidden_nonlinearity)      # tf.nn.tanh & co. by name
        output_nonlinearity = core.nonlinearity_name(output_nonlinearity)
        if hidden_nonlinearity not in self._activations or output_nonlinearity not in self._activations:
            raise ValueError("unsupported nonlinearity %r / %r (supported: %s)"
                             % (hidden_nonlinearity, output_nonlinearity, self._activations))
        self.cell_type = cell_type
        self.units = self.hidden_sizes[0]
        self.state_width = int(sum(self.hidden_sizes))
        self.hidden_nonlinearity = hidden_nonlinearity
        self.output_nonlinearity = output_nonlinearity
        self._spec = rnn_cells.param_spec(self.obs_space_dims, self.action_space_dims, self.hidden_sizes, cell_type)

        self._params = rnn_cells.init_params(self.obs_space_dims, self.action_space_dims, self.hidden_sizes,
                                             cell_type, np.random.RandomState(init_seed))
        self._native = None
        self._native_dirty = True

    # ------------------------------------------------------------------ parameters
    def get_param_values(self):
        return OrderedDict((k, p.numpy().copy()) for (k, _), p in zip(self._spec, self._params))

    def set_params(self, params):
        if isinstance(params, (dict, OrderedDict)):
            params = [params[k] for k, _ in self._spec]
        out = []
        for p in params:
            t = p.detach().clone() if torch.is_tensor(p) else torch.from_numpy(np.array(p, dtype=np.float32))
            out.append(t.to(dtype=torch.float32, device="cpu").contiguous())
        assert [tuple(t.shape) for t in out] == [shape for _, shape in self._spec], \
            "expected parameters %s" % (self._spec,)
        self._params = out
        self._native_dirty = True

    def set_normalization(self, normalization):
        self.normalization = normalization
        self._native_dirty = True

    def planner_model(self):
        from .native_lstm import NativeLSTM
        if self._native is None:
            self._native = NativeLSTM(self.obs_space_dims, self.action_space_dims, self.hidden_sizes,
                                      self.hidden_nonlinearity, self.output_nonlinearity, cell_type=self.cell_type)
            self._native_dirty = True
        if self._native_dirty:
            self._native.set_weights(self._params)
            if self.normalize_input:
                assert self.normalization is not None, "model has no normalization yet (call fit first)"
            self._native.set_norm(self.normalization if self.normalize_input else None)
            self._native_dirty = False
        return self._native

    def planner_blocks(self, m):
        """One shared weight set for every env (the planner interface of the MLP models)."""
        return 1

    # ------------------------------------------------------------------ hidden state (reference :273-293)
    def get_initial_hidden(self, batch_size):
        return rnn_cells.initial_hidden(self.cell_type, self.hidden_sizes, batch_size)

    def pack_hidden(self, hidden):
        """Reference hidden-state structure -> flat ``(c, h)`` float32 arrays ``[rows, sum(units)]``."""
        return rnn_cells.pack_hidden(self.cell_type, self.hidden_sizes, hidden)

    def unpack_hidden(self, c, h, as_tuple=False):
        return rnn_cells.unpack_hidden(self.cell_type, self.hidden_sizes, c, h, as_tuple=as_tuple)

    # ------------------------------------------------------------------ predict (reference :233-252)
    def predict(self, obs, act, hidden_state):
        assert obs.shape[0] == act.shape[0]
        assert obs.ndim == 2 and obs.shape[1] == self.obs_space_dims
        assert act.ndim == 2 and act.shape[1] == self.action_space_dims
        native = self.planner_model()
        c, h = self.pack_hidden(hidden_state)
        dev = native.device
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)  # noqa: E731
        nxt, c_out, h_out = native.predict(up(obs), up(act), up(c), up(h))
        pred_obs = nxt.cpu().numpy().astype(np.float64)
        return pred_obs, self.unpack_hidden(c_out.cpu().numpy(), h_out.cpu().numpy(), as_tuple=True)

    # ------------------------------------------------------------------ fit (reference :102-231)
    def compute_normalization(self, obs, act, obs_next):
        assert obs.shape[0] == obs_next.shape[0] == act.shape[0]
        assert obs.shape[1] == obs_next.shape[1] == act.shape[1]
        delta = obs_next - obs
        assert delta.ndim == 3 and delta.shape[2] == obs_next.shape[2] == obs.shape[2]
        norm = OrderedDict()
        norm["obs"] = (np.mean(obs, axis=(0, 1)), np.std(obs, axis=(0, 1)))
        norm["delta"] = (np.mean(delta, axis=(0, 1)), np.std(delta, axis=(0, 1)))
        norm["act"] = (np.mean(act, axis=(0, 1)), np.std(act, axis=(0, 1)))
        self.set_normalization(norm)

    def fit(self, obs, act, obs_next, epochs=1000, compute_normalization=True,
            valid_split_ratio=None, rolling_average_persitency=None, verbose=False, log_tabular=False):
        assert obs.ndim == 3 and obs.shape[2] == self.obs_space_dims
        assert obs_next.ndim == 3 and obs_next.shape[2] == self.obs_space_dims
        assert act.ndim == 3 and act.shape[2] == self.action_space_dims
        if valid_split_ratio is None:
            valid_split_ratio = self.valid_split_ratio
        if rolling_average_persitency is None:
            rolling_average_persitency = self.rolling_average_persitency
        assert 1 > valid_split_ratio >= 0

        if (self.normalization is None or compute_normalization) and self.normalize_input:
            self.compute_normalization(obs, act, obs_next)
        if self.normalize_input:
            nm = self.normalization
            obs_n = core.normalize(obs, nm["obs"][0], nm["obs"][1])
            act_n = core.normalize(act, nm["act"][0], nm["act"][1])
            delta_n = core.normalize(obs_next - obs, nm["delta"][0], nm["delta"][1])
        else:
            obs_n, act_n, delta_n = obs, act, obs_next - obs

        n_paths = obs_n.shape[0]                         # whole paths are split (:336-349)
        perm = np.arange(n_paths)
        np.random.shuffle(perm)
        split = int(n_paths * (1 - valid_split_ratio))
        tr, te = perm[:split], perm[split:]
        new_train = dict(obs=obs_n[tr], act=act_n[tr], delta=delta_n[tr])
        new_test = dict(obs=obs_n[te], act=act_n[te], delta=delta_n[te])
        if self._dataset_test is None:
            self._dataset_train, self._dataset_test = new_train, new_test
        else:
            for key in ("obs", "act", "delta"):
                self._dataset_train[key] = np.concatenate([self._dataset_train[key], new_train[key]])
                self._dataset_test[key] = np.concatenate([self._dataset_test[key], new_test[key]])

        dev = core.training_device()
        f32 = lambda a: torch.as_tensor(a, dtype=torch.float32, device=dev)  # noqa: E731
        x_tr = f32(np.concatenate([self._dataset_train["obs"], self._dataset_train["act"]], axis=2))
        y_tr = f32(self._dataset_train["delta"])
        x_te = f32(np.concatenate([self._dataset_test["obs"], self._dataset_test["act"]], axis=2))
        y_te = f32(self._dataset_test["delta"])

        params = [p.to(dev).requires_grad_(True) for p in self._params]
        opt = core.TFAdam(params, lr=self.learning_rate)
        rolling, rolling_prev = None, None
        epoch_times, last_epoch = [], 0
        zero_state = lambda b: rnn_cells.zero_state(self.cell_type, self.hidden_sizes, b, dev)  # noqa: E731
        forward = lambda x, st: rnn_cells.stack_forward(x, st, params, self.hidden_sizes, self.cell_type,  # noqa: E731
                                                        self.hidden_nonlinearity, self.output_nonlinearity)
        for epoch in range(epochs):
            t0 = time.time()
            # the reference batches first and shuffles the batches (:258-260)
            starts = list(range(0, x_tr.shape[0], self.batch_size))
            np.random.shuffle(starts)
            losses = []
            for s in starts:
                xb, yb = x_tr[s:s + self.batch_size], y_tr[s:s + self.batch_size]
                state = zero_state(xb.shape[0])
                sums, n_chunks = None, 0
                for i in range(0, xb.shape[1], self.backprop_steps):       # truncated BPTT (:165-180)
                    pred, state = forward(xb[:, i:i + self.backprop_steps], state)
                    loss = torch.mean((pred - yb[:, i:i + self.backprop_steps]) ** 2)
                    grads = torch.autograd.grad(loss, params)
                    state = rnn_cells.detach_state(state)
                    sums = list(grads) if sums is None else [a + g for a, g in zip(sums, grads)]
                    n_chunks += 1
                    losses.append(float(loss.detach()))
                for p, g in zip(params, sums):                              # mean over chunks (:182-184)
                    p.grad = g / n_chunks
                opt.step()
            with torch.no_grad():
                if x_te.shape[0] > 0:
                    vpred, _ = forward(x_te, zero_state(x_te.shape[0]))
                    valid_loss = float(torch.mean((vpred - y_te) ** 2))
                else:
                    valid_loss = float(np.mean(losses)) if losses else 0.0
            if rolling is None:
                rolling = 1.5 * valid_loss
                rolling_prev = 2 * valid_loss
            rolling = rolling_average_persitency * rolling + (1.0 - rolling_average_persitency) * valid_loss
            epoch_times.append(time.time() - t0)
            last_epoch = epoch
            if verbose:
                print("Training RNNDynamicsModel - finished epoch %i -- train loss: %.4f  valid loss: %.4f  "
                      "valid_loss_mov_avg: %.4f  epoch time: %.2f"
                      % (epoch, float(np.mean(losses)) if losses else float("nan"), valid_loss, rolling,
                         epoch_times[-1]))
            if rolling_prev < rolling or epoch == epochs - 1:
                break
            rolling_prev = rolling
        self._params = [p.detach().to("cpu").contiguous() for p in params]
        self._native_dirty = True
        self.fit_stats = dict(AvgModelEpochTime=float(np.mean(epoch_times)) if epoch_times else 0.0,
                              Epochs=last_epoch)
        return self.fit_stats

    # ------------------------------------------------------------------ pickling (reference :320-331)
    def __getstate__(self):
        state = dict()
        state["init_args"] = Serializable.__getstate__(self)
        state["normalization"] = self.normalization
        state["networks"] = [dict(network_params=self.get_param_values())]
        return state

    def __setstate__(self, state):
        Serializable.__setstate__(self, state["init_args"])
        self.normalization = state["normalization"]
        self.set_params(state["networks"][0]["network_params"])
