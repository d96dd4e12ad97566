"""``MetaMLPDynamicsModel`` - drop-in for ``learning_to_adapt/dynamics/meta_mlp_dynamics.py`` (GrBAL).

Hot-path part: after ``adapt`` the batch handed to ``predict`` is
split into equal row blocks and block *i* runs through adapted weight set *i*
(``meta_mlp_dynamics.py:296-306,143-163``).  The fused planner does the same with
``L2A_MODE_PER_BLOCK``: env *i* <-> weight set *i*; the adapted sets are uploaded ONCE per
``adapt`` call instead of being fed through ``feed_dict`` on every horizon step
(``:429-432``).

"Next" part (section 8(f) rank 1): ``adapt`` itself - one SGD step per env on its last
``adapt_batch_size`` transitions, ``theta_i = theta - alpha * grad MSE_i``
(``:321-345,409-421,106-120``) - is stock PyTorch autograd on the device, so the adapted
weights never visit the host.  Meta-training (``fit``, ``:165-275``) is first-order stock
PyTorch as well and is not part of the fused path.
"""

from collections import OrderedDict

import numpy as np
import torch

from ..utils.serializable import Serializable
from . import core


class _ResidentSets(object):
    """``_adapted_param_values`` when the adapted sets were produced on the device by
    ``l2a_model_adapt_sgd``: they live in the planner's model already; parameter tensors are copied out
    only if somebody asks for them (the reference keeps host dicts, ``meta_mlp_dynamics.py:344``)."""

    def __init__(self, native, count):
        self._native, self._count, self._cache = native, count, {}

    def __len__(self):
        return self._count

    def __getitem__(self, i):
        if not 0 <= i < self._count:
            raise IndexError(i)
        if i not in self._cache:
            self._cache[i] = self._native.get_weights(i)
        return self._cache[i]

    def __iter__(self):
        return (self[i] for i in range(self._count))


class MetaMLPDynamicsModel(Serializable):
    _activations = core.ACTIVATION_NAMES

    def __init__(self,
                 name,
                 env,
                 hidden_sizes=(512, 512),
                 meta_batch_size=10,
                 hidden_nonlinearity="relu",
                 output_nonlinearity=None,
                 batch_size=500,
                 learning_rate=0.001,
                 inner_learning_rate=0.1,
                 normalize_input=True,
                 optimizer=None,
                 valid_split_ratio=0.2,
                 rolling_average_persitency=0.99,
                 init_seed=None,
                 ):
        Serializable.quick_init(self, locals())

        self.normalization = None
        self.normalize_input = normalize_input
        self.meta_batch_size = meta_batch_size
        self.valid_split_ratio = valid_split_ratio
        self.rolling_average_persitency = rolling_average_persitency
        self.batch_size = batch_size
        self.learning_rate = learning_rate
        self.inner_learning_rate = inner_learning_rate
        self.name = name
        self._dataset_train = None
        self._dataset_test = None
        self._prev_params = None
        self._params_dev = None          # (list identity, device, device copies) of self._params
        self._adapted_stacked = None     # adapted sets stacked along a leading axis (batched adapt)
        self.use_native_adapt = True     # False: inner step with stock PyTorch autograd even on the GPU
        self._cuda_ok = None             # torch.cuda.is_available(), asked once (2 - 4 us a call, on every adapt otherwise)
        self._adapt_stage = None         # float64 arrays the sampler's lists are stacked into
        self._adapted_norm_of = None     # (native model, normalization) whose vectors are already uploaded
        self._adapted_param_values = None
        self._num_adapted_models = 0

        self.obs_space_dims = int(env.observation_space.shape[0])
        self.action_space_dims = int(env.action_space.shape[0])
        hidden_nonlinearity = core.nonlinearity_name(hidden_nonlinearity)      # tf.nn.tanh & co. by name
        output_nonlinearity = core.nonlinearity_name(output_nonlinearity)
        if hidden_nonlinearity not in self._activations or output_nonlinearity not in self._activations:
            raise ValueError("unsupported nonlinearity %r / %r" % (hidden_nonlinearity, output_nonlinearity))
        self.hidden_sizes = tuple(int(h) for h in hidden_sizes)
        self.hidden_nonlinearity = hidden_nonlinearity
        self.output_nonlinearity = output_nonlinearity

        sizes = (self.obs_space_dims + self.action_space_dims,) + self.hidden_sizes + (self.obs_space_dims,)
        self._params = core.xavier_params(sizes, np.random.RandomState(init_seed))
        self._native_base = None        # single-set handle (pre-adapt predictions)
        self._native_adapted = None     # per-block handle (post-adapt)
        self._base_dirty = True
        self._adapted_dirty = True

    # ------------------------------------------------------------------ parameters
    def get_param_values(self):
        names = core.param_names(len(self.hidden_sizes))
        return OrderedDict((k, p.detach().cpu().numpy().copy()) for k, p in zip(names, self._params))

    def set_params(self, params):
        self._params = core.as_param_list(params, len(self.hidden_sizes))
        self._base_dirty = True

    def set_normalization(self, normalization):
        self.normalization = normalization
        self._base_dirty = True
        self._adapted_dirty = True

    def set_adapted_params(self, param_sets):
        """Install externally computed adapted weight sets (list of OrderedDict / flat lists)."""
        self._adapted_param_values = [core.as_param_list(p, len(self.hidden_sizes)) for p in param_sets]
        self._adapted_stacked = None
        self._num_adapted_models = len(param_sets)
        self._adapted_dirty = True

    def _norm(self):
        if not self.normalize_input:
            return None
        assert self.normalization is not None, "model has no normalization yet (call fit first)"
        return self.normalization

    @property
    def mode(self):
        return "per_block" if self._adapted_param_values is not None else "single"

    def _device_params(self, dev):
        """Device-resident copy of the base parameters.  ``self._params`` is only ever replaced as a
        whole list, so its identity is the cache key; the controller step then runs no CPU tensor op
        (a CPU ``clone`` wakes torch's whole intra-op thread pool: measured 85 ms stalls on a 128-core host)."""
        dev = torch.device(dev)
        c = self._params_dev
        if c is None or c[0] is not self._params or c[1] != dev:
            self._params_dev = c = (self._params, dev, [p.to(dev) for p in self._params])
        return c[2]

    def planner_blocks(self, m):
        return self._num_adapted_models if self._adapted_param_values is not None else 1

    def planner_model(self):
        from .native_model import NativeModel
        if self._adapted_param_values is not None:
            k = self._num_adapted_models
            if self._native_adapted is None or self._native_adapted.n_sets != k:
                if self._native_adapted is not None:
                    self._native_adapted.close()
                self._native_adapted = NativeModel(self.obs_space_dims, self.action_space_dims,
                                                   self.hidden_sizes, self.hidden_nonlinearity,
                                                   self.output_nonlinearity, k, "per_block")
                self._adapted_dirty = True
            if self._adapted_dirty and isinstance(self._adapted_param_values, _ResidentSets):
                pass                                             # written in place by l2a_model_adapt_sgd
            elif self._adapted_dirty:
                if self._adapted_stacked is not None:            # batched adapt(): all sets in one strided call
                    self._native_adapted.set_weights_stacked(0, self._adapted_stacked)
                else:
                    for i in range(k):
                        self._native_adapted.set_weights(i, self._adapted_param_values[i])
            if self._adapted_dirty:
                c = self._adapted_norm_of       # normalisation vectors change only with fit / set_normalization
                if c is None or c[0] is not self._native_adapted or c[1] is not self.normalization:
                    for i in range(k):
                        self._native_adapted.set_norm(i, self._norm())
                    self._adapted_norm_of = (self._native_adapted, self.normalization)
                self._adapted_dirty = False
            return self._native_adapted
        if self._native_base is None:
            self._native_base = NativeModel(self.obs_space_dims, self.action_space_dims, self.hidden_sizes,
                                            self.hidden_nonlinearity, self.output_nonlinearity, 1, "single")
            self._base_dirty = True
        if self._base_dirty:
            self._native_base.set_weights(0, self._device_params(self._native_base.device))
            self._native_base.set_norm(0, self._norm())
            self._base_dirty = False
        return self._native_base

    # ------------------------------------------------------------------ predict (reference :276-306)
    def predict(self, obs, act):
        assert obs.shape[0] == act.shape[0]
        assert obs.ndim == 2 and obs.shape[1] == self.obs_space_dims
        assert act.ndim == 2 and act.shape[1] == self.action_space_dims
        native = self.planner_model()
        n_blocks = self.planner_blocks(None)
        assert obs.shape[0] % n_blocks == 0, "rows must split evenly over the adapted models"
        o = torch.from_numpy(np.ascontiguousarray(obs, dtype=np.float32)).to(native.device)
        a = torch.from_numpy(np.ascontiguousarray(act, dtype=np.float32)).to(native.device)
        nxt = native.predict(o, a, n_blocks=n_blocks)
        pred_obs = nxt.cpu().numpy().astype(np.float64)
        if not native.ctx.check_or_degrade():      # a tile-split launch lost its partner: unsplit now, run again
            pred_obs = native.predict(o, a, n_blocks=n_blocks).cpu().numpy().astype(np.float64)
            native.ctx.launch_status()
        return pred_obs

    # ------------------------------------------------------------------ adapt (reference :321-351)
    def adapt(self, obs, act, obs_next):
        """One inner SGD step per env.  ``obs`` / ``act`` / ``obs_next``: lists of m arrays
        ``[adapt_batch_size, dim]`` (``samplers/sampler.py:81-90``)."""
        self._num_adapted_models = len(obs)
        assert len(obs) == len(act) == len(obs_next)
        xs = ys = None
        shape0 = np.shape(obs[0])
        uniform = all(np.shape(o) == shape0 for o in obs)
        if (uniform and self.normalize_input and len(shape0) == 2 and 1 <= shape0[0] <= 16 and self._native_adapt_ok(None)
                and self.obs_space_dims + self.action_space_dims <= 128):
            # the sampler's case on a GPU: raw float64 transitions straight to the library, which normalises them on
            # the device with the host's own arithmetic (l2a_model_adapt_sgd_raw) - the host only stacks and copies
            # (into arrays kept from step to step: this preamble sits in front of the step's first launch)
            m, rows = len(obs), shape0[0]
            st = self._adapt_stage
            if st is None or st[0].shape[:2] != (m, rows):
                st = self._adapt_stage = (np.empty((m, rows, self.obs_space_dims)), np.empty((m, rows, self.action_space_dims)),
                                          np.empty((m, rows, self.obs_space_dims)))
            ob, ac, ob_next = st
            assert shape0[1] == self.obs_space_dims
            np.concatenate(obs, out=ob.reshape(m * rows, -1))               # raises on any other shape / a mismatch between the lists
            np.concatenate(act, out=ac.reshape(m * rows, -1))
            np.concatenate(obs_next, out=ob_next.reshape(m * rows, -1))
            native = self._adapted_handle(m)
            native.adapt_sgd_raw(self._device_params(native.device), ob, ac, ob_next, self._norm(),
                                 self.inner_learning_rate)
            self._adapted_stacked = None
            self._prev_params = self._params      # parameter tensors are never modified in place
            self._adapted_param_values = _ResidentSets(native, m)
            self._adapted_dirty = True
            return
        dev = core.training_device()
        if uniform:
            # equal batches (the sampler's case): all tasks normalised in one pass - the same float64 arithmetic per
            # element as the per-task loop below, a dozen NumPy calls instead of thirty (0.1 ms of a 2.7 ms step)
            ob = np.asarray(obs, dtype=np.float64)
            ac = np.asarray(act, dtype=np.float64)
            ob_next = np.asarray(obs_next, dtype=np.float64)
            assert ob.ndim == 3 and ob.shape[2] == self.obs_space_dims
            assert ac.ndim == 3 and ac.shape[2] == self.action_space_dims
            assert ob_next.shape == ob.shape and ac.shape[:2] == ob.shape[:2]
            if self.normalize_input:
                nm = self._norm()
                o_n = core.normalize(ob, nm["obs"][0], nm["obs"][1])
                a_n = core.normalize(ac, nm["act"][0], nm["act"][1])
                d_n = core.normalize(ob_next - ob, nm["delta"][0], nm["delta"][1])
            else:
                o_n, a_n, d_n = ob, ac, ob_next - ob
            xs = np.concatenate([o_n, a_n], axis=2)         # [m, rows, in]: iterating yields the per-task arrays
            ys = d_n
        else:
            xs, ys = [], []
            for ob, ac, ob_next in zip(obs, act, obs_next):
                ob = np.asarray(ob, dtype=np.float64)
                ac = np.asarray(ac, dtype=np.float64)
                ob_next = np.asarray(ob_next, dtype=np.float64)
                assert ob.ndim == 2 and ob.shape[1] == self.obs_space_dims
                assert ac.ndim == 2 and ac.shape[1] == self.action_space_dims
                assert ob_next.shape == ob.shape and ac.shape[0] == ob.shape[0]
                if self.normalize_input:
                    nm = self._norm()
                    o_n = core.normalize(ob, nm["obs"][0], nm["obs"][1])
                    a_n = core.normalize(ac, nm["act"][0], nm["act"][1])
                    d_n = core.normalize(ob_next - ob, nm["delta"][0], nm["delta"][1])
                else:
                    o_n, a_n, d_n = ob, ac, ob_next - ob
                xs.append(np.concatenate([o_n, a_n], axis=1))
                ys.append(d_n)
        # Only the real rows enter the pre-update loss: the reference pads each task with an equal
        # number of zero rows and then splits the task batch in two, pre = real half (:324-326, :99-103).
        if self._native_adapt_ok(xs):
            # on the GPU: kernels that write the adapted sets straight into the planner's per-block model
            m = len(xs)
            native = self._adapted_handle(m)
            # the batches stay on the host: the library stages them in host-mapped memory that the kernels read
            # directly (l2a_model_adapt_sgd_host) - no H2D copies on the stream
            native.adapt_sgd_host(self._device_params(native.device), np.asarray(xs, dtype=np.float32),
                                  np.asarray(ys, dtype=np.float32), self.inner_learning_rate)
            self._adapted_stacked = None
            adapted = _ResidentSets(native, m)
        elif len({x.shape[0] for x in xs}) == 1:
            # all envs in ONE batched forward/backward: every env gets its own copy of theta, so the
            # gradient of sum_i L_i w.r.t. copy i is exactly grad L_i (:409-421)
            m = len(xs)
            x = torch.from_numpy(np.stack(xs).astype(np.float32)).to(dev)              # [m, rows, in]
            y = torch.from_numpy(np.stack(ys).astype(np.float32)).to(dev)
            params = [p.unsqueeze(0).expand((m,) + tuple(p.shape)).clone().requires_grad_(True)
                      for p in self._device_params(dev)]
            hid, out = core.torch_act(self.hidden_nonlinearity), core.torch_act(self.output_nonlinearity)
            n_layers = len(params) // 2
            t = x
            for li in range(n_layers):
                t = torch.baddbmm(params[2 * li + 1].unsqueeze(1), t, params[2 * li])
                t = hid(t) if li < n_layers - 1 else out(t)
            loss = torch.mean((y - t) ** 2, dim=(1, 2)).sum()                          # sum_i L_i (:118)
            grads = torch.autograd.grad(loss, params)
            stacked = [(p - self.inner_learning_rate * g).detach() for p, g in zip(params, grads)]
            adapted = [[q[i] for q in stacked] for i in range(m)]
            self._adapted_stacked = stacked
        else:
            self._adapted_stacked = None
            base = self._device_params(dev)
            adapted = []
            for x_np, y_np in zip(xs, ys):
                x = torch.from_numpy(x_np.astype(np.float32)).to(dev)
                y = torch.from_numpy(y_np.astype(np.float32)).to(dev)
                params = [p.detach().clone().requires_grad_(True) for p in base]
                pred = core.mlp_forward(x, params, self.hidden_nonlinearity, self.output_nonlinearity)
                loss = torch.mean((y - pred) ** 2)                                      # :118
                grads = torch.autograd.grad(loss, params)
                adapted.append([(p - self.inner_learning_rate * g).detach() for p, g in zip(params, grads)])  # :415-417
        self._prev_params = self._params      # parameter tensors are never modified in place
        self._adapted_param_values = adapted
        self._adapted_dirty = True

    def _adapted_handle(self, m):
        """The per-block planner handle the device adaptation writes its m sets into."""
        from .native_model import NativeModel
        if self._native_adapted is None or self._native_adapted.n_sets != m:
            if self._native_adapted is not None:
                self._native_adapted.close()
            self._native_adapted = NativeModel(self.obs_space_dims, self.action_space_dims, self.hidden_sizes,
                                               self.hidden_nonlinearity, self.output_nonlinearity, m, "per_block")
        return self._native_adapted

    def _native_adapt_ok(self, xs):
        """The fused device path needs a GPU, equal batches of at most 16 rows (``xs`` = None: the caller has checked
        that), an identity output layer and a hidden nonlinearity whose derivative follows from its output."""
        if not self.use_native_adapt:
            return False
        if self._cuda_ok is None:
            self._cuda_ok = bool(torch.cuda.is_available())
        if not self._cuda_ok:
            return False
        if xs is not None:
            rows = {x.shape[0] for x in xs}
            if not (len(rows) == 1 and 1 <= next(iter(rows)) <= 16):
                return False
        return (self.output_nonlinearity in (None, "identity")
                and self.hidden_nonlinearity in (None, "identity", "relu", "tanh", "sigmoid"))

    def switch_to_pre_adapt(self):
        if self._prev_params is not None:
            self._params = self._prev_params
            self._prev_params = None
            self._adapted_param_values = None
            self._base_dirty = True

    # ------------------------------------------------------------------ normalisation / fit
    def compute_normalization(self, obs, act, obs_next):
        """``obs`` etc. are ``[paths, path_len, dim]`` as in the reference (``:392-407``)."""
        assert obs.shape[:2] == obs_next.shape[:2] == act.shape[:2]
        delta = obs_next - obs
        norm = OrderedDict()
        norm["obs"] = (np.mean(obs, axis=(0, 1)), np.std(obs, axis=(0, 1)))
        norm["delta"] = (np.mean(delta, axis=(0, 1)), np.std(delta, axis=(0, 1)))
        norm["act"] = (np.mean(act, axis=(0, 1)), np.std(act, axis=(0, 1)))
        self.set_normalization(norm)

    def fit(self, obs, act, obs_next, epochs=1000, compute_normalization=True,
            valid_split_ratio=None, rolling_average_persitency=None, verbose=False, log_tabular=False):
        """Meta-training (stock PyTorch), mirroring the reference's ``fit`` (``:167-268``):

        * the paths of THIS call are normalised, split into train / validation by ``valid_split_ratio``
          (``train_test_split``, ``:453-466``: a shuffle of the path indices from the global NumPy generator) and
          APPENDED to ``_dataset_train`` / ``_dataset_test`` (``:193-203``) - the trainer hands over only the newest
          iteration's rollouts and relies on the model to keep the older ones;
        * a step samples ``meta_batch_size`` windows of ``2 * batch_size`` consecutive transitions from the
          accumulated train set (``_get_batch``, ``:353-390``), adapts on the first half of each window and minimises
          the mean post-update loss on the second half, differentiating through the inner step (``:96-141``);
        * after every epoch the plain (un-adapted) loss ``self.loss`` (``:88``) is averaged over windows sampled from
          the held-out set and drives the rolling-average early stop (``:236-262``)."""
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
            o_n = core.normalize(obs, nm["obs"][0], nm["obs"][1])
            a_n = core.normalize(act, nm["act"][0], nm["act"][1])
            d_n = core.normalize(obs_next - obs, nm["delta"][0], nm["delta"][1])
        else:
            o_n, a_n, d_n = obs, act, obs_next - obs

        # train_test_split (:453-466) over the paths, then accumulate (:193-203)
        n_paths = o_n.shape[0]
        indices = np.arange(n_paths)
        np.random.shuffle(indices)
        split_idx = int(n_paths * (1 - valid_split_ratio))
        idx_train, idx_test = indices[:split_idx], indices[split_idx:]
        new_train = dict(obs=o_n[idx_train], act=a_n[idx_train], delta=d_n[idx_train])
        new_test = dict(obs=o_n[idx_test], act=a_n[idx_test], delta=d_n[idx_test])
        if self._dataset_test is None:
            self._dataset_test, self._dataset_train = new_test, new_train
        else:
            for key in ("obs", "act", "delta"):
                self._dataset_test[key] = np.concatenate([self._dataset_test[key], new_test[key]])
                self._dataset_train[key] = np.concatenate([self._dataset_train[key], new_train[key]])

        dev = core.training_device()

        def to_dev(ds):
            x = torch.as_tensor(np.concatenate([ds["obs"], ds["act"]], axis=2), dtype=torch.float32, device=dev)
            return x, torch.as_tensor(ds["delta"], dtype=torch.float32, device=dev)
        X, Y = to_dev(self._dataset_train)
        # a held-out set can be empty (valid_split_ratio = 0, or a single path): validate on the train set then
        # (the reference would fail in np.random.randint(0, 0))
        XT, YT = to_dev(self._dataset_test) if self._dataset_test["obs"].shape[0] > 0 else (X, Y)
        bs, mbs = self.batch_size, self.meta_batch_size
        assert X.shape[1] > 2 * bs, "paths must be longer than 2 * batch_size"
        params = [p.to(dev).requires_grad_(True) for p in self._params]
        opt = core.TFAdam(params, lr=self.learning_rate)
        steps_per_epoch = max(int(np.prod(X.shape[:2]) / (mbs * bs * 2)), 1)                   # :209-210
        steps_test = max(int(np.prod(XT.shape[:2]) / (mbs * bs * 2)), 1)                       # :211-212
        hid, out = self.hidden_nonlinearity, self.output_nonlinearity

        def windows(x, y):          # _get_batch (:353-390): meta_batch_size windows of 2 * batch_size transitions
            num_paths, len_path = x.shape[:2]
            ip = np.random.randint(0, num_paths, size=mbs)
            ib = np.random.randint(bs, len_path - bs, size=mbs)
            return [(x[p_i, b_i - bs:b_i + bs], y[p_i, b_i - bs:b_i + bs]) for p_i, b_i in zip(ip, ib)]

        rolling, rolling_prev, last_epoch = None, None, 0
        pre_losses, post_losses = [], []
        for epoch in range(epochs):
            pre_losses, post_losses = [], []
            for _ in range(steps_per_epoch):
                pre_total, post_total = 0.0, 0.0
                for xw, yw in windows(X, Y):
                    xa, ya, xb, yb = xw[:bs], yw[:bs], xw[bs:], yw[bs:]                       # pre / post halves :99-100
                    pre = torch.mean((ya - core.mlp_forward(xa, params, hid, out)) ** 2)       # :118
                    grads = torch.autograd.grad(pre, params, create_graph=True)                # :409-421
                    fast = [p - self.inner_learning_rate * g for p, g in zip(params, grads)]
                    post_total = post_total + torch.mean((yb - core.mlp_forward(xb, fast, hid, out)) ** 2)   # :133
                    pre_total = pre_total + pre.detach()
                loss = post_total / mbs                                                        # :139
                opt.zero_grad(set_to_none=True)
                loss.backward()
                opt.step()
                post_losses.append(float(loss.detach()))
                pre_losses.append(float(pre_total / mbs))
            with torch.no_grad():                                                              # :227-236
                valid_losses = []
                for _ in range(steps_test):
                    ws = windows(XT, YT)
                    xv = torch.cat([w[0] for w in ws], dim=0)
                    yv = torch.cat([w[1] for w in ws], dim=0)
                    valid_losses.append(float(torch.mean((yv - core.mlp_forward(xv, params, hid, out)) ** 2)))   # :88
            valid_loss = float(np.mean(valid_losses))
            if rolling is None:                                                                # :238-243
                rolling, rolling_prev = 1.5 * valid_loss, 2 * valid_loss
                if valid_loss < 0:
                    rolling, rolling_prev = valid_loss / 1.5, valid_loss / 2
            rolling = rolling_average_persitency * rolling + (1.0 - rolling_average_persitency) * valid_loss
            last_epoch = epoch
            if verbose:
                print("Training MetaDynamicsModel - epoch %i -- train loss: %.4f  valid loss: %.4f  mov_avg: %.4f"
                      % (epoch, float(np.mean(post_losses)), valid_loss, rolling))
            if rolling_prev < rolling or epoch == epochs - 1:                                  # :258-261
                break
            rolling_prev = rolling
        self._params = [p.detach().to("cpu").contiguous() for p in params]
        self._base_dirty = True
        self.fit_stats = {"Epochs": last_epoch, "Post-Loss": float(np.mean(post_losses)) if post_losses else None,
                          "Pre-Loss": float(np.mean(pre_losses)) if pre_losses else None,
                          "TrainPaths": int(X.shape[0]), "ValidPaths": int(self._dataset_test["obs"].shape[0])}
        return self.fit_stats

    # ------------------------------------------------------------------ pickling (reference :434-445)
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
