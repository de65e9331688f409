"""Shared by the task drivers: the repo root on sys.path (they are run as `python tasks/taskNN.py`, like the reference's) and the reference's
argument parsers (tasks/task02.py:15-22, tasks/task04.py / task05.py:34-43 of nbip/IWAE)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

_EPOCHS_HELP = ("numper of epochs, if set to -1 number of epochs "
                "will be set based on the learning rate scheme from the paper")


def parser_task02():
    p = argparse.ArgumentParser()
    p.add_argument("--n_samples", type=int, default=5, help="number of importance samples")
    p.add_argument("--batch_size", type=int, default=20, help="batch size")
    p.add_argument("--epochs", type=int, default=-1, help=_EPOCHS_HELP)
    p.add_argument("--gpu", type=str, default='0', help="Choose GPU")
    return p


def parser_conditional():
    p = argparse.ArgumentParser()
    p.add_argument("--stochastic_layers", type=int, default=1, choices=[1, 2], help="number of stochastic layers in the model")
    p.add_argument("--n_samples", type=int, default=5, help="number of importance samples")
    p.add_argument("--batch_size", type=int, default=20, help="batch size")
    p.add_argument("--epochs", type=int, default=-1, help=_EPOCHS_HELP)
    p.add_argument("--objective", type=str, default="iwae_elbo", choices=["vae_elbo", "iwae_elbo", "iwae_eq14", "vae_elbo_kl"])
    p.add_argument("--gpu", type=str, default='0', help="Choose GPU")
    return p
