"""ALSOption / BPRMFOption / WARPOption / CFROption / EALSOption: key names, defaults and validation are part of the ABI
(the backends parse the JSON dump of these dicts) -- /root/reference/buffalo/algo/options.py:4-311.
Only deviation: `accelerator` defaults to True because this package has no CPU backend."""
from ..misc import InputOptions, Option


class AlgoOption(InputOptions):
    def get_default_option(self):  # options.py:8-31
        return {
            "evaluation_on_learning": True, "compute_loss_on_training": True, "early_stopping_rounds": 0,
            "save_best": False, "evaluation_period": 1, "save_period": 10, "random_seed": 0, "validation": {},
        }

    def is_valid_option(self, opt):  # options.py:33-38
        b = super().is_valid_option(opt)
        if "num_workers" not in opt:
            raise RuntimeError("num_workers not defined")
        return b


class ALSOption(AlgoOption):
    def get_default_option(self):  # options.py:44-86
        opt = super().get_default_option()
        opt.update({
            "adaptive_reg": False, "save_factors": False, "accelerator": True, "d": 20, "num_iters": 10,
            "num_workers": 1, "hyper_threads": 256, "num_cg_max_iters": 3, "reg_u": 0.1, "reg_i": 0.1,
            "alpha": 8.0, "optimizer": "manual_cg", "cg_tolerance": 1e-10, "block_size": 32, "eps": 1e-10,
            "model_path": "", "data_opt": {},
        })
        return Option(opt)

    def is_valid_option(self, opt):  # options.py:88-95
        b = super().is_valid_option(opt)
        possible = ["llt", "ldlt", "manual_cg", "eigen_cg", "eigen_bicg", "eigen_gmres", "eigen_dgmres",
                    "eigen_minres", "ialspp"]
        if opt.optimizer not in possible:
            raise RuntimeError(f"optimizer ({opt.optimizer}) should be in {possible}")
        return b


class BPRMFOption(AlgoOption):
    def get_default_option(self):  # options.py:193-252
        opt = super().get_default_option()
        opt.update({
            "accelerator": True, "use_bias": True, "evaluation_period": 100, "num_workers": 1,
            "hyper_threads": 256, "num_iters": 100, "d": 20, "update_i": True, "update_j": True,
            "reg_u": 0.025, "reg_i": 0.025, "reg_j": 0.025, "reg_b": 0.025, "optimizer": "sgd", "lr": 0.002,
            "min_lr": 0.0001, "beta1": 0.9, "beta2": 0.999, "eps": 1e-10, "per_coordinate_normalize": False,
            "num_negative_samples": 1, "sampling_power": 0.0, "verify_neg": True, "random_positive": False,
            "model_path": "", "data_opt": {},
        })
        return Option(opt)


class WARPOption(AlgoOption):
    def get_default_option(self):  # options.py:260-311
        opt = super().get_default_option()
        opt.update({
            "accelerator": True, "evaluation_period": 5, "num_workers": 1, "hyper_threads": 256, "num_iters": 40,
            "d": 64, "threshold": 1.0, "score_func": "dot", "max_trials": 500, "update_i": True, "update_j": True,
            "reg_u": 0.0, "reg_i": 0.0, "reg_j": 0.0, "optimizer": "adagrad", "lr": 0.05, "min_lr": 0.0001,
            "beta1": 0.9, "beta2": 0.999, "eps": 1e-10, "per_coordinate_normalize": False, "model_path": "",
            "data_opt": {},
        })
        return Option(opt)


class EALSOption(AlgoOption):
    def get_default_option(self):  # options.py:102-132
        opt = super().get_default_option()
        opt.update({
            "save_factors": False, "d": 20, "num_iters": 10, "num_workers": 1, "reg_u": 0.1, "reg_i": 0.1, "alpha": 8.0,
            "c0": 512.0, "exponent": 0.5, "model_path": "", "data_opt": {},
        })
        return Option(opt)


class CFROption(AlgoOption):
    def get_default_option(self):  # options.py:139-177
        opt = super().get_default_option()
        opt.update({
            "save_factors": False, "d": 20, "num_iters": 10, "num_workers": 1, "num_cg_max_iters": 3, "cg_tolerance": 1e-10,
            "eps": 1e-10, "reg_u": 0.1, "reg_i": 0.1, "reg_c": 0.1, "alpha": 8.0, "l": 1.0, "optimizer": "manual_cg",
            "model_path": "", "data_opt": {},
        })
        return Option(opt)

    def is_valid_option(self, opt):  # options.py:179-186
        b = super().is_valid_option(opt)
        possible = ["llt", "ldlt", "manual_cg", "eigen_cg", "eigen_bicg", "eigen_gmres", "eigen_dgmres", "eigen_minres"]
        if opt.optimizer not in possible:
            raise RuntimeError(f"optimizer ({opt.optimizer}) should be in {possible}")
        return b
