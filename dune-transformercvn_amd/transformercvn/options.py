"""Option bag of the TransformerCVN hot path.

Drop-in for the reference's ``transformercvn.options.Options`` (reference: transformercvn/options.py:7-188): a flat
``argparse.Namespace`` whose attributes are overridden from a JSON option file with int/bool coercion
(options.py:164-173); unknown JSON keys are added as-is.  Defaults are kept in one table below.
"""
from __future__ import annotations

import json
from argparse import Namespace
from typing import Any, Dict

# name -> default  (grouped as in the reference file; values from options.py:21-162)
_DEFAULTS: Dict[str, Any] = {
    # network architecture
    "hidden_dim": 128, "initial_feature_dim": 32, "initial_pixel_dim": 16,
    "feature_embedding_dim": 8, "pixel_embedding_dim": 512, "position_embedding_dim": 16,
    "final_decoder_dim": 16, "num_embedding_layers": 100, "num_encoder_layers": 5, "num_decoder_layers": 100,
    "num_prong_decoder_layers": 4, "num_attention_heads": 8, "transformer_activation": "gelu",
    "transformer_norm_first": False, "linear_prelu_activation": True, "linear_batch_norm": True,
    "disable_smart_features": False, "normalize_features": True, "one_hot_pixels": False, "log_pixels": False,
    "mobilenet_structure": None, "densenet_structure": [6, 12, 24, 16], "densenet_growth_rate": 16,
    "densenet_batch_norm_size": 4,
    # dataset
    "dataset_limit": 1.0, "train_validation_split": 0.95, "batch_size": 2048, "num_dataloader_workers": 8,
    "load_full_dataset": False, "event_current_targets": False,
    # training
    "optimizer": "AdamW", "learning_rate": 0.0001, "l2_penalty": 0.015, "gradient_clip": 90.0, "dropout": 0.0,
    "epochs": 25, "learning_rate_warmup_epochs": 1.0, "learning_rate_cycles": 1, "num_gpu": 1,
    "event_prong_loss_proportion": 0.5, "loss_beta": 2.5, "loss_gamma": 0.0, "pixel_noise_std": 0.01,
    # misc
    "verbose_output": True, "usable_gpus": "", "trial_time": "", "trial_output_dir": "./test_output",
}


class Options(Namespace):
    def __init__(self, training_file: str = "", testing_file: str = "", validation_file: str = ""):
        super().__init__()
        for key, value in _DEFAULTS.items():
            setattr(self, key, list(value) if isinstance(value, list) else value)
        self.training_file = training_file
        self.testing_file = testing_file
        self.validation_file = validation_file

    def update_options(self, new_options: Dict[str, Any]) -> None:
        """JSON overlay: keys whose current value is an int (bools included) are coerced with int(), bool keys with
        bool() -- same precedence as the reference (int check first, options.py:165-171)."""
        current = vars(self)
        as_int = {k for k, v in current.items() if isinstance(v, int)}
        as_bool = {k for k, v in current.items() if isinstance(v, bool)}
        for key, value in new_options.items():
            if key in as_int:
                value = int(value)
            elif key in as_bool:
                value = bool(value)
            setattr(self, key, value)

    @classmethod
    def load(cls, filepath: str) -> "Options":
        options = cls()
        with open(filepath, "r") as handle:
            options.update_options(json.load(handle))
        return options

    def display(self) -> None:
        bar = "=" * 70
        print(bar + "\nOptions\n" + "-" * 70)
        for key in sorted(vars(self)):
            print(f"{key:32}: {getattr(self, key)}")
        print(bar)
