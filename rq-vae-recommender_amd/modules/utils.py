"""Small helpers with the reference's names (reference modules/utils.py:7-22)."""
import argparse
import functools

try:  # the real gin-config when present, the in-tree subset otherwise
    import gin
except ImportError:  # pragma: no cover - depends on the environment
    from rqhip import ginlite as gin


def eval_mode(fn):
    """Run a module method with `self.eval()`, restoring the previous training flag afterwards."""
    @functools.wraps(fn)
    def inner(self, *args, **kwargs):
        was_training = self.training
        self.eval()
        try:
            return fn(self, *args, **kwargs)
        finally:
            self.train(was_training)
    return inner


def parse_config(argv=None) -> None:
    """`python train_rqvae.py <config.gin>`: one positional argument, parsed into gin bindings."""
    parser = argparse.ArgumentParser()
    parser.add_argument("config_path", type=str, help="Path to gin config file.")
    args = parser.parse_args(argv)
    gin.parse_config_file(args.config_path)
