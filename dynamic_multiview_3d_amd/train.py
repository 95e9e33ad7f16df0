"""Train driver -- drop-in for dyn_mult_view/multi_view_model/train.py:34-156.

Same flags (--hyper, --visualize, --device, --pretrained), same conf files (a python module with a
`configuration` dict; reference-style confs that `from appearance_flow_model import ...` load
unchanged), same model selection (conf['model'], default Base_Prediction_Model: train.py:57-60), same
loop cadence (iterations itr_0..num_iterations inclusive, log every 10, validation every 500,
checkpoint every 10 000 to output_dir/model<itr> as a TensorFlow V2 bundle (tf_checkpoint.py), resume iteration
parsed from the checkpoint name: train.py:95-103,117-154).  sess.run([loss, train_op]) is model.train_step().

Input: the TFRecord shards under conf['data_dir'] (read_tf_records.py); when that directory holds no files, or with
--synthetic, seeded synthetic batches shaped like the reader's tensors.

--visualize <checkpoint name> restores output_dir/<name> and writes the model's qualitative outputs (visualize.py).
Not ported: TF summaries (a JSON-lines log is written instead).
"""
import argparse
import importlib
import importlib.util
import json
import os
import sys
import time
import types

import numpy as np
import torch

SUMMARY_INTERVAL = 400      # train.py:24
VAL_INTERVAL = 500          # train.py:27
SAVE_INTERVAL = 10000       # train.py:30

_ALIASES = ('appearance_flow_model', 'highdim_angle', 'lowdim_angle', 'appearance_flow_tinghui', 'main_model',
            'multiobject_appflow', 'multiobject_main_model')


def load_conf(conf_file):
    """imp.load_source('hyperparams', conf_file).configuration (train.py:44-45), with the reference's
    bare module names resolved to this package."""
    if not os.path.exists(conf_file):
        sys.exit("Experiment configuration not found")
    from . import compat
    compat.install()                               # bare model-module names and dyn_mult_view.* (confs use dyn_mult_view.__file__)
    spec = importlib.util.spec_from_file_location('hyperparams', conf_file)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.configuration


def select_model(conf):
    if 'model' in conf:
        return conf['model']
    from .main_model import Base_Prediction_Model
    return Base_Prediction_Model


class SyntheticData:
    """Seeded stand-in for build_tfrecord_input (multi_view_model/utils/read_tf_records.py:15-85):
    uint8-quantised car-like renders / 255, masks in {0,1}, displacement ranges of the datasets
    (SURVEY 8d).  A small pool of device-resident batches is cycled."""

    def __init__(self, model, seed=0, pool=4):
        self.model = model
        rng = np.random.default_rng(seed)
        self.pool = []
        for _ in range(pool):
            batch = {}
            for name, t in model.graph.inputs.items():
                if len(t.shape) == 2:
                    if name == 'displacement':
                        a = rng.normal(10, 10, t.shape) * rng.choice([-1, 1], t.shape)
                    else:
                        a = np.stack([rng.uniform(-1, 1, t.shape[0]), rng.uniform(-6.28, 6.28, t.shape[0])], 1)
                else:
                    a = self._images(rng, t.shape)
                    if 'mask' in name:
                        a = (a[..., :1] > 0.55).astype(np.float32) * np.ones(t.shape, np.float32)
                batch[name] = torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).to(model.graph.device)
            self.pool.append(batch)
        self.i = 0

    @staticmethod
    def _images(rng, shape):
        b, h, w, c = shape
        yy, xx = np.mgrid[0:h, 0:w]
        img = np.full(shape, 127.0, np.float32)
        for i in range(b):
            cy, cx = rng.uniform(0.3 * h, 0.7 * h, 2)
            ay, ax = rng.uniform(0.12 * h, 0.35 * h, 2)
            img[i][((yy - cy) / ay) ** 2 + ((xx - cx) / ax) ** 2 <= 1] = rng.uniform(0, 255, c)
        img += rng.normal(0, 2, shape).astype(np.float32)
        return np.clip(np.rint(img), 0, 255) / 255.0

    def next(self):
        b = self.pool[self.i % len(self.pool)]
        self.i += 1
        return b


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split('\n')[0])
    ap.add_argument('--hyper', default='../../tensorflowdata/appflow_offset/conf.py', help='hyperparameters configuration file')
    ap.add_argument('--visualize', default='', help='model within hyperparameter folder from which to create gifs')
    ap.add_argument('--device', default='0', help='GPU index (the reference sets CUDA_VISIBLE_DEVICES)')
    ap.add_argument('--pretrained', default=None, help='path to model file from which to resume training')
    ap.add_argument('--num_iterations', type=int, default=None, help='override conf["num_iterations"]')
    ap.add_argument('--synthetic', action='store_true', help='ignore conf["data_dir"] and train on synthetic batches')
    FLAGS = ap.parse_args(argv)

    conf = load_conf(FLAGS.hyper)
    if FLAGS.visualize:                                               # train.py:47-55
        print('creating visualizations ...')
        conf['data_dir'] = '/'.join(str.split(conf.get('data_dir') or '', '/')[:-1] + ['test'])
        conf['visualize'] = conf['output_dir'] + '/' + FLAGS.visualize
        conf['event_log_dir'] = '/tmp'
        conf['batch_size'] = 10
        conf['test_mode'] = ''
    if FLAGS.num_iterations is not None:
        conf['num_iterations'] = FLAGS.num_iterations

    from . import parallel
    rank, world, local_rank = parallel.init_from_env()
    dev = 'cuda:%d' % (local_rank if world > 1 else int(FLAGS.device))
    torch.cuda.set_device(torch.device(dev))

    Model = select_model(conf)
    model = Model(conf, load_tfrec=True, build_loss=not FLAGS.visualize, device=dev)
    if world > 1:
        comm = parallel.make_comm(rank, world, 'rccl' if torch.cuda.is_available() else 'gloo')
        model.enable_data_parallel(world, comm=comm)
    saver = model.saver
    data_dir = conf.get('data_dir')
    if not FLAGS.synthetic and data_dir and os.path.isdir(data_dir) and os.listdir(data_dir):
        # the reference's shards (multi_view_model/utils/read_tf_records.py:15-85), read without TensorFlow
        from .read_tf_records import build_tfrecord_input
        train_data = build_tfrecord_input(conf, model, training=True, seed=rank, rank=rank, world=world)
        val_data = build_tfrecord_input(conf, model, training=False, seed=10_000 + rank)
        if rank == 0:
            print('reading TFRecord shards from', data_dir)
    else:
        if rank == 0:
            print('no TFRecord shards at conf["data_dir"] = %r: training on synthetic batches' % (data_dir,))
        train_data = SyntheticData(model, seed=rank)
        val_data = SyntheticData(model, seed=10_000 + rank, pool=1)

    if FLAGS.visualize:                                               # train.py:80-92
        print('-------------------------------------------------------------------')
        print('verify current settings!! ')
        for key in conf.keys():
            print(key, ': ', conf[key])
        print('-------------------------------------------------------------------')
        saver.restore(None, conf['visualize'])
        print('restore done.')
        model.visualize(None, **train_data.next())
        return model

    itr_0 = 0
    if FLAGS.pretrained is not None:
        conf['pretrained_model'] = FLAGS.pretrained
        saver.restore(None, conf['pretrained_model'])
        from .model_base import iteration_from_checkpoint_name
        itr_0 = iteration_from_checkpoint_name(conf['pretrained_model'])      # train.py:99-101
        print('resuming training at iteration:  ', itr_0)

    if rank == 0:
        print('-------------------------------------------------------------------')
        print('verify current settings!! ')
        for key in conf.keys():
            print(key, ': ', conf[key])
        print('-------------------------------------------------------------------')
        os.makedirs(conf['output_dir'], exist_ok=True)
        log = open(os.path.join(conf['output_dir'], 'train_log.jsonl'), 'a')

    # The launch thread runs tens of milliseconds ahead of the GPU; a full cyclic-GC pass over everything the imports and
    # the graph construction left behind takes longer than that and drains the queues.  Move those objects to the
    # permanent generation: later collections only look at what the loop itself allocates.
    import gc
    gc.collect()
    gc.freeze()

    starttime = time.time()
    t_iter = []
    for itr in range(itr_0, conf['num_iterations'] + 1, 1):         # inclusive, train.py:117
        t_startiter = time.time()
        cost = model.train_step(**train_data.next())
        if itr % 10 == 0:
            c = float(cost)
            if rank == 0:
                print(str(itr) + ' ' + str(c))
                log.write(json.dumps({'itr': itr, 'training_loss': c}) + '\n')
        if itr % VAL_INTERVAL == 0 and itr != 0:
            vc = float(model.forward(**val_data.next()))
            if rank == 0:
                log.write(json.dumps({'itr': itr, 'val_loss': vc}) + '\n')
        if itr % SAVE_INTERVAL == 0 and itr != 0:
            model.graph.gather_optimizer_state()        # collective: the sharded optimiser's slots, complete on every rank
            if rank == 0:
                print('Saving model to' + conf['output_dir'])
                saver.save(None, conf['output_dir'] + '/model' + str(itr))
        t_iter.append(time.time() - t_startiter)
        if itr % 100 == 1 and rank == 0:
            torch.cuda.synchronize()
            avg_t_iter = (time.time() - starttime) / (itr - itr_0 + 1)
            print('time per iteration: {0}'.format(avg_t_iter))
            print('expected for complete training: {0}h '.format(avg_t_iter / 3600 * conf['num_iterations']))
            log.flush()
    model.graph.gather_optimizer_state()
    if rank == 0:
        print('Saving model.')
        saver.save(None, conf['output_dir'] + '/model')
        log.close()
        print('Training complete')
    return model


if __name__ == '__main__':
    main()
