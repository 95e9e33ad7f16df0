"""Shared machinery of the model classes: graph ownership, feeding, stepping, checkpoints.

The reference classes own a TF graph + placeholders and are driven by
sess.run([model.loss, model.train_op], {model.train_cond: 1}) (multi_view_model/train.py:120-123).
Here `train_step(**feeds)` / `forward(**feeds)` are the explicit replacements of those
sess.run calls; everything else keeps the reference attribute names.
"""
import os
import re

import numpy as np
import torch

from . import tf_checkpoint
from .graph import Graph


class Saver:
    """tf.train.Saver stand-in (train.py:70-71,134-136; mv3d/utils/tf_utils.py:199-212): variables + Adam slots +
    beta powers under their TF names, written as a TensorFlow V2 checkpoint (`<prefix>.index`,
    `<prefix>.data-00000-of-00001`, and the `checkpoint` state file next to them) -- see tf_checkpoint.py."""

    def __init__(self, graph):
        self.graph = graph

    def save(self, sess, save_path, global_step=None):
        prefix = save_path if global_step is None else '%s-%d' % (save_path, int(global_step))
        prefix = os.path.abspath(prefix)
        tf_checkpoint.write_checkpoint(prefix, {k: v.numpy() for k, v in self.graph.state_dict().items()})
        tf_checkpoint.update_checkpoint_state(os.path.dirname(prefix), prefix)
        return prefix

    def restore(self, sess, save_path):
        if tf_checkpoint.checkpoint_exists(save_path):
            arrays = tf_checkpoint.read_checkpoint(save_path)
            self.graph.load_state_dict({k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in arrays.items()})
        elif os.path.isfile(save_path):                      # state dict written by torch.save (pre-bundle snapshots)
            self.graph.load_state_dict(torch.load(save_path, map_location='cpu', weights_only=True))      # tensors only: never unpickle objects
        else:
            raise FileNotFoundError("no checkpoint at %r (expected %s.index)" % (save_path, save_path))


class AdamOptimizer:
    """tf.train.AdamOptimizer(lr).minimize(loss) (appearance_flow_model.py:77): TF defaults
    beta1=0.9, beta2=0.999, epsilon=1e-8; the update itself is mv3d_adam_step."""

    def __init__(self, learning_rate, beta1=0.9, beta2=0.999, epsilon=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = learning_rate, beta1, beta2, epsilon

    def minimize(self, loss, graph):
        graph.loss_expr = loss
        graph.lr = self.lr
        graph.beta1, graph.beta2, graph.eps = self.beta1, self.beta2, self.eps
        return 'train_op'


class ModelBase(object):
    input_names = ()

    def _make_graph(self, device, seed):
        self.graph = Graph(device=device, seed=seed)
        return self.graph

    def _finish(self, build_loss):
        self.t_vars = list(self.graph.variables.keys())
        self.saver = Saver(self.graph)
        self.graph.compile()

    # ---- sess.run replacements
    def feed(self, **feeds):
        for k, v in feeds.items():
            if v is None:
                continue
            if k not in self.graph.inputs:
                raise KeyError("unknown input %r (have %s)" % (k, list(self.graph.inputs)))
            self.graph.inputs[k].set(v)

    def train_step(self, **feeds):
        """One sess.run([model.loss, model.train_op], {train_cond: 1}); returns the loss as a
        0-d device tensor (call float() on it to synchronise)."""
        if self.graph.loss_expr is None:
            raise RuntimeError("model was built with build_loss=False")
        self.feed(**feeds)
        return self.graph.train_step()

    def forward(self, **feeds):
        """One forward pass (the train_cond: 0 / visualize path, train.py:130, appearance_flow_model.py:134)."""
        self.feed(**feeds)
        self.graph.run_forward()
        return self.graph.loss_buf[0]

    # ---- data-parallel hook
    def enable_data_parallel(self, world_size, group=None, comm=None, mode=None):
        """comm: parallel.RcclComm (the product path: RCCL through the C ABI) or parallel.TorchComm (default: torch.distributed
        on `group`; gloo in the CPU rehearsals).  mode: 'sharded' (default) or 'allreduce' -- Graph.run_backward_overlapped."""
        from .parallel import TorchComm
        g = self.graph
        g.world_size = int(world_size)
        g.dist_group = group
        g.comm = comm if comm is not None else TorchComm(group)
        if mode is not None:
            g.dp_mode = mode
        g.upload_adam_state()          # gradient scale 1 / world size for the SUM


def iteration_from_checkpoint_name(path):
    """train.py:99-101: resume iteration = trailing digits of the checkpoint file name."""
    return int(re.match('.*?([0-9]+)$', path).group(1))
