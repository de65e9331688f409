"""Driver of the reference's tasks/task02.py (:110-259): the 1-layer IWAE trained with the doubly reparameterised gradient estimator (DReG),
same flags (--n_samples --batch_size --epochs --gpu), same schedule and evaluation protocol, on the MI355X-native step (objective IWAE_OBJ_DREG,
BASELINE configs[3]).  The training loop is main.py's (run_training); plots are out of scope.

    python tasks/task02.py --n_samples 50 --batch_size 20
"""
from _common import parser_task02

from iwae_amd import task02
from main import run_training


def main(argv=None):
    args = parser_task02().parse_args(argv)
    string = "task02_{0}".format(args.n_samples)                       # tasks/task02.py:112
    return run_training(args, string, lambda **kw: task02.IWAEDReG(200, 100, **kw), "dreg", report="iwae_elbo")


if __name__ == "__main__":
    main()
