"""``MLPDynamicsModel`` - drop-in for ``learning_to_adapt/dynamics/mlp_dynamics.py:11-222``.

Same constructor keywords as the reference (including the mis-spelt
``rolling_average_persitency`` and the nonlinearity given as the *string* ``'relu'``,
``run_scripts/run_mb_mpc.py:23-32``), same ``predict`` / ``fit`` / ``normalization`` surface.
New keyword ``ensemble_size`` (default 1 = the reference's single network): E independently
initialised networks whose predicted deltas are averaged (BASELINE.json's ``ens=5``).

* ``predict`` (reference ``:204-222``) runs on the MI355X through ``l2a_predict``.
* The planner does not call ``predict`` per horizon step; ``MPCController`` asks the model for
  its ``NativeModel`` handle (``planner_model()``) and launches the fused rollout.
* ``fit`` (reference ``:91-202``) is stock PyTorch-ROCm (Adam, shuffled mini-batches, the same
  rolling-average early stop); training is not part of the fused hot path.
"""

import time
from collections import OrderedDict

import numpy as np
import torch

from ..utils.serializable import Serializable
from . import core


class MLPDynamicsModel(Serializable):
    """Feed-forward model of normalised state deltas."""

    _activations = core.ACTIVATION_NAMES

    def __init__(self,
                 name,
                 env,
                 hidden_sizes=(512, 512),
                 hidden_nonlinearity="relu",
                 output_nonlinearity=None,
                 batch_size=500,
                 learning_rate=0.001,
                 normalize_input=True,
                 optimizer=None,
                 valid_split_ratio=0.2,
                 rolling_average_persitency=0.99,
                 ensemble_size=1,
                 init_seed=None,
                 ):
        Serializable.quick_init(self, locals())

        self.normalization = None
        self.normalize_input = normalize_input
        self.valid_split_ratio = valid_split_ratio
        self.rolling_average_persitency = rolling_average_persitency
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.name = name
        self._dataset_train = None
        self._dataset_test = None

        # determine dimensionality of state and action space (reference :54-55)
        self.obs_space_dims = int(env.observation_space.shape[0])
        self.action_space_dims = int(env.action_space.shape[0])

        hidden_nonlinearity = core.nonlinearity_name(hidden_nonlinearity)      # tf.nn.tanh & co. by name
        output_nonlinearity = core.nonlinearity_name(output_nonlinearity)
        if hidden_nonlinearity not in self._activations or output_nonlinearity not in self._activations:
            raise ValueError("unsupported nonlinearity %r / %r (supported: %s)"
                             % (hidden_nonlinearity, output_nonlinearity, self._activations))
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        self.hidden_nonlinearity = hidden_nonlinearity
        self.output_nonlinearity = output_nonlinearity
        self.ensemble_size = int(ensemble_size)
        assert self.ensemble_size >= 1

        sizes = (self.obs_space_dims + self.action_space_dims,) + self.hidden_sizes + (self.obs_space_dims,)
        rng = np.random.RandomState(init_seed)
        self._param_sets = [core.xavier_params(sizes, rng) for _ in range(self.ensemble_size)]
        self._norm_sets = None          # optional per-member normalisation (default: shared)
        self._native = None
        self._native_dirty = True

    # ------------------------------------------------------------------ parameters
    @property
    def mode(self):
        return "single" if self.ensemble_size == 1 else "mean"

    def get_param_values(self, member=0):
        """The reference's ``network_params`` OrderedDict (``dynamics/core/layers.py:71-79``)."""
        names = core.param_names(len(self.hidden_sizes))
        return OrderedDict((k, p.numpy().copy()) for k, p in zip(names, self._param_sets[member]))

    def set_params(self, params, member=0):
        """``params``: OrderedDict name -> array (reference ``layers.py:81-94``) or flat list."""
        self._param_sets[member] = core.as_param_list(params, len(self.hidden_sizes))
        self._native_dirty = True

    def set_normalization(self, normalization, per_member=None):
        """``normalization``: the reference's dict.  ``per_member``: optional list of E dicts."""
        self.normalization = normalization
        self._norm_sets = per_member
        self._native_dirty = True

    def _norm_of(self, e):
        if not self.normalize_input:
            return None
        if self._norm_sets is not None:
            return self._norm_sets[e]
        assert self.normalization is not None, "model has no normalization yet (call fit first)"
        return self.normalization

    def planner_model(self):
        """Return the up-to-date ``NativeModel`` (creates / refreshes the HBM copy lazily)."""
        from .native_model import NativeModel
        if self._native is None:
            self._native = NativeModel(self.obs_space_dims, self.action_space_dims, self.hidden_sizes,
                                       self.hidden_nonlinearity, self.output_nonlinearity,
                                       self.ensemble_size, self.mode)
            self._native_dirty = True
        if self._native_dirty:
            for e in range(self.ensemble_size):
                self._native.set_weights(e, self._param_sets[e])
                self._native.set_norm(e, self._norm_of(e))
            self._native_dirty = False
        return self._native

    def planner_blocks(self, m):
        """How many weight-set blocks the planner must pass for ``m`` envs (1 = shared)."""
        return 1

    # ------------------------------------------------------------------ predict (reference :204-222)
    def predict(self, obs, act):
        assert obs.shape[0] == act.shape[0]
        assert obs.ndim == 2 and obs.shape[1] == self.obs_space_dims
        assert act.ndim == 2 and act.shape[1] == self.action_space_dims
        native = self.planner_model()
        o = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float32)).to(native.device)
        a = torch.from_numpy(np.ascontiguousarray(act, dtype=np.float32)).to(native.device)
        nxt = native.predict(o, a)
        pred_obs = nxt.cpu().numpy().astype(np.float64)
        if not native.ctx.check_or_degrade():      # a tile-split launch lost its partner: unsplit now, run again
            pred_obs = native.predict(o, a).cpu().numpy().astype(np.float64)
            native.ctx.launch_status()
        assert pred_obs.ndim == 2
        return pred_obs

    # ------------------------------------------------------------------ fit (reference :91-202)
    def compute_normalization(self, obs, act, obs_next):
        assert obs.shape[0] == obs_next.shape[0] == act.shape[0]
        delta = obs_next - obs
        assert delta.ndim == 2 and delta.shape[0] == obs_next.shape[0]
        norm = OrderedDict()
        norm["obs"] = (np.mean(obs, axis=0), np.std(obs, axis=0))
        norm["delta"] = (np.mean(delta, axis=0), np.std(delta, axis=0))
        norm["act"] = (np.mean(act, axis=0), np.std(act, axis=0))
        self.set_normalization(norm)

    def _normalize_data(self, obs, act, obs_next=None):
        nm = self.normalization
        obs_n = core.normalize(obs, nm["obs"][0], nm["obs"][1])
        act_n = core.normalize(act, nm["act"][0], nm["act"][1])
        if obs_next is not None:
            delta_n = core.normalize(obs_next - obs, nm["delta"][0], nm["delta"][1])
            return obs_n, act_n, delta_n
        return obs_n, act_n

    def fit(self, obs, act, obs_next, epochs=1000, compute_normalization=True,
            valid_split_ratio=None, rolling_average_persitency=None, verbose=False, log_tabular=False):
        assert obs.ndim == 2 and obs.shape[1] == self.obs_space_dims
        assert obs_next.ndim == 2 and obs_next.shape[1] == self.obs_space_dims
        assert act.ndim == 2 and act.shape[1] == self.action_space_dims
        if valid_split_ratio is None:
            valid_split_ratio = self.valid_split_ratio
        if rolling_average_persitency is None:
            rolling_average_persitency = self.rolling_average_persitency
        assert 1 > valid_split_ratio >= 0

        if (self.normalization is None or compute_normalization) and self.normalize_input:
            self.compute_normalization(obs, act, obs_next)
        if self.normalize_input:
            obs_n, act_n, delta_n = self._normalize_data(obs, act, obs_next)
        else:
            obs_n, act_n, delta_n = obs, act, obs_next - obs

        # train / validation split (reference train_test_split, :273-285)
        n_data = obs_n.shape[0]
        perm = np.arange(n_data)
        np.random.shuffle(perm)
        split = int(n_data * (1 - valid_split_ratio))
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
        x_tr = torch.as_tensor(np.concatenate([self._dataset_train["obs"], self._dataset_train["act"]], axis=1),
                               dtype=torch.float32, device=dev)
        y_tr = torch.as_tensor(self._dataset_train["delta"], dtype=torch.float32, device=dev)
        x_te = torch.as_tensor(np.concatenate([self._dataset_test["obs"], self._dataset_test["act"]], axis=1),
                               dtype=torch.float32, device=dev)
        y_te = torch.as_tensor(self._dataset_test["delta"], dtype=torch.float32, device=dev)

        epoch_times, last_epoch = [], 0
        for e in range(self.ensemble_size):
            params = [p.to(dev).requires_grad_(True) for p in self._param_sets[e]]
            opt = core.TFAdam(params, lr=self.learning_rate)
            rolling, rolling_prev = None, None
            for epoch in range(epochs):
                t0 = time.time()
                order = torch.randperm(x_tr.shape[0], device=dev)
                losses = []
                for s in range(0, x_tr.shape[0], self.batch_size):
                    idx = order[s:s + self.batch_size]
                    pred = core.mlp_forward(x_tr[idx], params, self.hidden_nonlinearity, self.output_nonlinearity)
                    loss = torch.mean((y_tr[idx] - pred) ** 2)
                    opt.zero_grad(set_to_none=True)
                    loss.backward()
                    opt.step()
                    losses.append(float(loss.detach()))
                with torch.no_grad():
                    if x_te.shape[0] > 0:
                        vpred = core.mlp_forward(x_te, params, self.hidden_nonlinearity, self.output_nonlinearity)
                        valid_loss = float(torch.mean((y_te - vpred) ** 2))
                    else:
                        valid_loss = float(np.mean(losses)) if losses else 0.0
                if rolling is None:
                    rolling = 1.5 * valid_loss
                    rolling_prev = 2 * valid_loss
                rolling = rolling_average_persitency * rolling + (1.0 - rolling_average_persitency) * valid_loss
                epoch_times.append(time.time() - t0)
                last_epoch = epoch
                if verbose:
                    print("Training DynamicsModel[%d] - epoch %i -- train loss: %.4f  valid loss: %.4f  "
                          "valid_loss_mov_avg: %.4f  epoch time: %.2f"
                          % (e, epoch, float(np.mean(losses)) if losses else float("nan"), valid_loss,
                             rolling, epoch_times[-1]))
                if rolling_prev < rolling or epoch == epochs - 1:
                    break
                rolling_prev = rolling
            self._param_sets[e] = [p.detach().to("cpu").contiguous() for p in params]
        self._native_dirty = True
        self.fit_stats = dict(AvgModelEpochTime=float(np.mean(epoch_times)) if epoch_times else 0.0,
                              Epochs=last_epoch)
        return self.fit_stats

    # ------------------------------------------------------------------ pickling
    def __getstate__(self):
        # The reference class defines no __getstate__, so its weights are lost on snapshot
        # (SURVEY.md section 5).  Here they are kept, in the reference's `network_params` format.
        state = dict()
        state["init_args"] = Serializable.__getstate__(self)
        state["normalization"] = self.normalization
        state["norm_sets"] = self._norm_sets
        state["networks"] = [dict(network_params=self.get_param_values(e)) for e in range(self.ensemble_size)]
        return state

    def __setstate__(self, state):
        Serializable.__setstate__(self, state["init_args"])
        self.normalization = state["normalization"]
        self._norm_sets = state.get("norm_sets")
        for e, net in enumerate(state["networks"]):
            self.set_params(net["network_params"], member=e)
