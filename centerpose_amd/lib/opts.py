"""Flag system of the reference (``/root/reference/src/lib/opts.py``), mirrored for the inference path.

Same flag names, types and defaults (``opts.__init__`` :15-328), the same derived fields
(``parse`` :330-376) and dataset constants / head dictionary (``update_dataset_info_and_set_heads``
:378-429, ``init`` :431-502), so the entry scripts' ``opts().parser.parse_args()`` ->
``opts().parse(opt)`` -> ``opts().init(opt)`` sequence (demo.py:93-155) works unchanged.  The flag
table is data, not code; defaults are pinned against the reference parser by
tests/golden/opts_defaults.json.
"""
import argparse
import os

_CATS = ('bike', 'book', 'bottle', 'camera', 'cereal_box', 'chair', 'cup', 'mug', 'laptop', 'shoe')

# (name, kind, default[, type])   kind: 'flag' = store_true, 'val' = valued option
_FLAGS = [
    # basic experiment setting
    ('task', 'val', 'object_pose'), ('dataset', 'val', 'objectron'), ('exp_id', 'val', 'default'),
    ('test', 'flag'), ('debug', 'val', 1, int), ('demo', 'val', ''), ('show_axes', 'flag'),
    ('demo_save', 'val', '../demo/'), ('load_model', 'val', ''), ('resume', 'flag'),
    # system
    ('gpus', 'val', '0'), ('num_workers', 'val', 4, int), ('not_cuda_benchmark', 'flag'), ('seed', 'val', 317, int),
    # log
    ('print_iter', 'val', 0, int), ('hide_data_time', 'flag'), ('save_all', 'flag'), ('metric', 'val', 'loss'),
    ('vis_thresh', 'val', 0.3, float), ('debugger_theme', 'choice', 'white', ['white', 'black']),
    ('paper_display', 'flag'),
    # model
    ('arch', 'val', 'dla_34'), ('head_conv', 'val', -1, int), ('down_ratio', 'val', 4, int),
    # input
    ('input_res', 'val', -1, int), ('input_h', 'val', -1, int), ('input_w', 'val', -1, int),
    # train
    ('lr', 'val', 1.25e-4, float), ('lr_step', 'val', '90,120', str), ('num_epochs', 'val', 20, int),
    ('batch_size', 'val', 32, int), ('master_batch_size', 'val', -1, int), ('num_iters', 'val', -1, int),
    ('val_intervals', 'val', 5, int), ('trainval', 'flag'),
    # test
    ('test_scales', 'val', '1', str), ('nms', 'flag'), ('K', 'val', 100, int), ('not_prefetch_test', 'flag'),
    ('fix_res', 'flag'), ('fix_short', 'val', -1, int), ('keep_res', 'flag'),
    # dataset
    ('not_rand_crop', 'flag'), ('shift', 'val', 0.05, float), ('scale', 'val', 0.4, float), ('rotate', 'val', 0, float),
    ('flip', 'val', 0.5, float), ('no_color_aug', 'flag'), ('aug_rot', 'val', 0, float),
    # loss
    ('mse_loss', 'flag'), ('reg_loss', 'val', 'l1'), ('hm_weight', 'val', 1, float), ('off_weight', 'val', 1, float),
    ('wh_weight', 'val', 0.1, float), ('hp_weight', 'val', 1, float), ('hm_hp_weight', 'val', 1, float),
    # task
    ('not_reg_offset', 'flag'), ('center_thresh', 'val', 0.3, float), ('dense_hp', 'flag'), ('not_hm_hp', 'flag'),
    ('not_reg_hp_offset', 'flag'), ('not_reg_bbox', 'flag'),
    # object pose
    ('c', 'val', 'chair'), ('hps_uncertainty', 'flag'), ('obj_scale', 'flag'), ('obj_scale_uncertainty', 'flag'),
    ('obj_scale_weight', 'val', 1, float), ('use_pnp', 'flag'), ('mug', 'flag'), ('num_symmetry', 'val', 12),
    ('cam_intrinsic', 'nargs', None, float), ('rep_mode', 'val', 1, int), ('data_generation_mode_ratio', 'val', 0, float),
    ('center_3D', 'flag'), ('use_residual', 'flag'), ('use_absolute_scale', 'flag'), ('new_data_augmentation', 'flag'),
    ('balance_coefficient', 'val', {k: 2 for k in _CATS}), ('conf_border', 'val', {k: [3, 9] for k in _CATS}),
    ('R', 'val', 20, float),
    # tracking
    ('refined_Kalman', 'flag'), ('tracking_task', 'flag'), ('tracking', 'flag'), ('tracking_hp', 'flag'),
    ('pre_hm', 'flag'), ('pre_hm_hp', 'flag'), ('same_aug_pre', 'flag'), ('hm_heat_random', 'flag'),
    ('hm_disturb', 'val', 0, float), ('lost_disturb', 'val', 0, float), ('fp_disturb', 'val', 0, float),
    ('hm_hp_heat_random', 'flag'), ('hm_hp_disturb', 'val', 0, float), ('hp_lost_disturb', 'val', 0, float),
    ('hp_fp_disturb', 'val', 0, float), ('KL_scale_uncertainty', 'val', 0.1, float),
    ('KL_kps_uncertainty', 'val', 0.1, float), ('tracking_label_mode', 'val', 1), ('render_hm_mode', 'val', 1),
    ('render_hmhp_mode', 'val', 2), ('pre_thresh', 'val', -1, float), ('track_thresh', 'val', 0.3, float),
    ('new_thresh', 'val', 0.3, float), ('max_frame_dist', 'val', 3, int), ('pre_img', 'flag'), ('hungarian', 'flag'),
    ('kalman', 'flag'), ('scale_pool', 'flag'), ('max_age', 'val', 5, int), ('tracking_weight', 'val', 1, float),
    ('tracking_hp_weight', 'val', 0.5, float), ('gt_pre_hm_hmhp', 'flag'), ('gt_pre_hm_hmhp_first', 'flag'),
    ('empty_pre_hm', 'flag'),
    # ground-truth substitution switches of the evaluator
    ('eval_oracle_hm', 'flag'), ('eval_oracle_wh', 'flag'), ('eval_oracle_offset', 'flag'), ('eval_oracle_kps', 'flag'),
    ('eval_oracle_hmhp', 'flag'), ('eval_oracle_hp_offset', 'flag'), ('eval_oracle_dep', 'flag'),
]

# dataset constants of opts.init (:433-498)
_OBJECT_POSE_INFO = {
    'default_resolution': [512, 512], 'num_classes': 1,
    'mean': [0.408, 0.447, 0.470], 'std': [0.289, 0.274, 0.278],
    'dataset': 'objectron', 'num_joints': 8,
    'flip_idx': [[1, 5], [3, 7], [2, 6], [4, 8]],
    'dimension_ref': {
        'bike': [[0.65320896, 1.021797894, 1.519635599, 0.6520559199, 1.506392621],
                 [0.1179380561, 0.176747817, 0.2981715678, 0.1667947895, 0.3830536275]],
        'book': [[0.225618019, 0.03949624326, 0.1625821624, 7.021850281, 5.064694187],
                 [0.1687487664, 0.07391230822, 0.06436673199, 3.59629568, 2.723290812]],
        'bottle': [[0.07889784977450116, 0.24127451915330908, 0.0723714257114412, 0.33644069262302545,
                    0.3091134992864717],
                   [0.02984649578071775, 0.06381390122918497, 0.03088144838560917, 0.11052240441921059,
                    0.13327627592012867]],
        'camera': [[0.11989848700326843, 0.08226238775595619, 0.09871718158089632, 1.507216484439368,
                    1.1569407159290284],
                   [0.021177290310316968, 0.02158788017191602, 0.055673710278419844, 0.28789183678046854,
                    0.5342094080365904]],
        'cereal_box': [[0.19202754401417296, 0.2593114001714919, 0.07723794925413519, 0.7542602699204104,
                        0.29441151268928173],
                       [0.08481640897407464, 0.09999915952084068, 0.09495429981036707, 0.19829004029411457,
                        0.2744797990483879]],
        'chair': [[0.5740664085137888, 0.8434027515832329, 0.6051523831888338, 0.6949691013776601,
                   0.7326891354260606],
                  [0.12853104253707456, 0.14852086453095492, 0.13428881418587957, 0.16897092539619352,
                   0.18636134566748525]],
        'cup': [[0.08587637391801063, 0.12025228955138188, 0.08486836104868696, 0.7812126934904675,
                 0.7697895244331658],
                [0.05886805978497525, 0.06794896438246326, 0.05875681990718713, 0.2887038681446475,
                 0.283821205157399]],
        'mug': [[0.14799136566553112, 0.09729087667918128, 0.08845449667169905, 1.3875694883045138,
                 1.0224997119392225],
                [1.0488828523223728, 0.2552672927963539, 0.039095350310480705, 0.3947832854104711,
                 0.31089415283872546]],
        'laptop': [[0.33685059747485196, 0.1528068814247063, 0.2781020624738614, 35.920214652427696,
                    23.941173992376903],
                   [0.03529983948867832, 0.07017080198389423, 0.0665823136876069, 391.915687801732,
                    254.21325950495455]],
        'shoe': [[0.10308848289662519, 0.10932616184503478, 0.2611737789760352, 1.0301976264129833,
                  2.6157393112424328],
                 [0.02274768925924402, 0.044958380226590516, 0.04589720205423542, 0.3271000267177176,
                  0.8460337534776092]],
    },
}


def _csv(text, kind):
    return [kind(v) for v in text.split(',')]


def _gpu_ids(text):
    ids = _csv(text, int)
    return list(range(len(ids))) if ids[0] >= 0 else [-1]


def _chunks(batch, master, n_gpus):
    """Per-GPU chunk sizes of a DataParallel batch: the master's share first, the rest spread as evenly as it divides."""
    rest, others = batch - master, n_gpus - 1
    return [master] + [rest // others + (1 if i < rest % others else 0) for i in range(others)]



class opts(object):
    def __init__(self):
        self.parser = argparse.ArgumentParser()
        for f in _FLAGS:
            name, kind = '--' + f[0], f[1]
            if kind == 'flag':
                self.parser.add_argument(name, action='store_true')
            elif kind == 'choice':
                self.parser.add_argument(name, default=f[2], choices=f[3])
            elif kind == 'nargs':
                self.parser.add_argument(name, default=f[2], nargs='+', type=f[3])
            elif len(f) > 3:
                self.parser.add_argument(name, default=f[2], type=f[3])
            else:
                self.parser.add_argument(name, default=f[2])

    # ---- derived options: one table, evaluated in order (each rule sees the fields set before it).  The field names, their
    #      meaning and the two console lines are the reference's interface (opts.py:330-429); the formulation is this repo's,
    #      and tests/golden/opts_scenarios.json pins the resulting Namespace field by field against the reference's own parse().
    _DERIVED = (
        ('gpus_str', lambda o: o.gpus),
        ('gpus', lambda o: _gpu_ids(o.gpus)),
        ('lr_step', lambda o: _csv(o.lr_step, int)),
        ('test_scales', lambda o: _csv(o.test_scales, float)),
        ('fix_res', lambda o: not o.keep_res),
        ('reg_offset', lambda o: not o.not_reg_offset),
        ('reg_bbox', lambda o: not o.not_reg_bbox),
        ('hm_hp', lambda o: not o.not_hm_hp),
        ('reg_hp_offset', lambda o: o.hm_hp and not o.not_reg_hp_offset),
        ('head_conv', lambda o: o.head_conv if o.head_conv != -1 else (256 if 'dla' in o.arch else 64)),
        ('pad', lambda o: 127 if 'hourglass' in o.arch else 31),
        ('num_stacks', lambda o: 2 if o.arch == 'hourglass' else 1),
        ('val_intervals', lambda o: 100000000 if o.trainval else o.val_intervals),
        # --debug > 0: single worker-less, single-image, single-GPU run
        ('num_workers', lambda o: 0 if o.debug > 0 else o.num_workers),
        ('batch_size', lambda o: 1 if o.debug > 0 else o.batch_size),
        ('gpus', lambda o: o.gpus[:1] if o.debug > 0 else o.gpus),
        ('master_batch_size', lambda o: o.batch_size // len(o.gpus) if (o.debug > 0 or o.master_batch_size == -1)
            else o.master_batch_size),
        ('chunk_sizes', lambda o: _chunks(o.batch_size, o.master_batch_size, len(o.gpus))),
        ('root_dir', lambda o: os.path.join(os.path.dirname(__file__), '..', '..')),
        ('data_dir', lambda o: os.path.join(o.root_dir, 'data')),
        ('exp_dir', lambda o: os.path.join(o.root_dir, 'exp', o.task)),
        ('save_dir', lambda o: os.path.join(o.exp_dir, o.exp_id)),
        ('debug_dir', lambda o: os.path.join(o.save_dir, 'debug')),
    )
    # head name -> (channels, enabled?) in the order the reference inserts them into the dict
    _HEADS = (
        ('hm', lambda o: o.num_classes, lambda o: True),
        ('wh', lambda o: 2, lambda o: True),
        ('hps', lambda o: 16, lambda o: True),
        ('hps_uncertainty', lambda o: 16, lambda o: o.hps_uncertainty),
        ('reg', lambda o: 2, lambda o: o.reg_offset),
        ('hm_hp', lambda o: 8, lambda o: o.hm_hp),
        ('hp_offset', lambda o: 2, lambda o: o.reg_hp_offset),
        ('scale', lambda o: 3, lambda o: o.obj_scale),
        ('scale_uncertainty', lambda o: 3, lambda o: o.obj_scale and o.obj_scale_uncertainty),
        ('tracking', lambda o: 2, lambda o: o.tracking == True),        # noqa: E712 (the reference's truthiness test)
        ('tracking_hp', lambda o: 16, lambda o: o.tracking_hp == True),  # noqa: E712
    )

    def parse(self, opt):
        """Derived options (the reference's opts.py:330-376 as a table)."""
        for name, rule in self._DERIVED:
            setattr(opt, name, rule(opt))
        print('Fix size testing.' if opt.fix_res else 'Keep resolution testing.')
        print('training chunk_sizes:', opt.chunk_sizes)
        print('The output will be saved to ', opt.save_dir)
        return opt

    def update_dataset_info_and_set_heads(self, opt, dataset):
        """Resolution fields and the head dictionary (opts.py:378-429)."""
        opt.mean, opt.std, opt.num_classes, opt.flip_idx = dataset.mean, dataset.std, dataset.num_classes, dataset.flip_idx
        for axis, default in zip(('input_h', 'input_w'), dataset.default_resolution):
            explicit = getattr(opt, axis)
            setattr(opt, axis, explicit if explicit > 0 else (opt.input_res if opt.input_res > 0 else default))
        opt.output_h, opt.output_w = opt.input_h // opt.down_ratio, opt.input_w // opt.down_ratio
        opt.input_res, opt.output_res = max(opt.input_h, opt.input_w), max(opt.output_h, opt.output_w)
        opt.heads = {name: width(opt) for name, width, on in self._HEADS if on(opt)}
        if opt.use_residual:
            ref = dataset.dimension_ref['mug' if (opt.c == 'cup' and opt.mug) else opt.c][0]
            opt.dimension_ref = ref[0:3] if opt.use_absolute_scale else [ref[3], 1, ref[4]]
        print('heads', opt.heads)
        return opt

    def init(self, opt):
        """opts.py:431-502"""
        class Struct:
            def __init__(self, entries):
                for k, v in entries.items():
                    self.__setattr__(k, v)

        info = {'object_pose': _OBJECT_POSE_INFO}
        dataset = Struct(info[opt.task])
        opt.dataset = dataset.dataset
        return self.update_dataset_info_and_set_heads(opt, dataset)
