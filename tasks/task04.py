"""Driver of the reference's tasks/task04.py (:205-355): the conditional IWAE with a LEARNED conditional prior p(z|y) (a third BasicBlock on
onehot(y), tasks/task04.py:108,124-130), trained on (x, y) batches; same flags, same loop as tasks/task05.py here.

    python tasks/task04.py --n_samples 5 --batch_size 20 --objective iwae_elbo
"""
from _common import parser_conditional  # noqa: F401  (puts the repo root on sys.path)

from iwae_amd import task04
import task05 as _t5


def main(argv=None):
    return _t5.main(argv, module=task04, name="task04")


if __name__ == "__main__":
    main()
