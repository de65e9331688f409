"""Driver of the reference's tasks/task05.py (:200-345): the label-conditional IWAE (CIWAE: encoder on concat(x, onehot(y)), decoder on
concat(z, onehot(y)), prior N(0, 1)) trained on (x, y) batches, same flags as the reference's script.  The labelled training set stays resident in
HBM (iwae_dataset_upload + iwae_dataset_set_labels); the loop is main.py's (run_training).  --stochastic_layers 2 selects the unconditional 2-layer
model exactly as the reference's script does (tasks/task05.py:255-258); plots are out of scope.

    python tasks/task05.py --n_samples 5 --batch_size 20 --objective iwae_elbo
"""
from _common import parser_conditional

from iwae_amd import iwae2, task05
from main import run_training


def main(argv=None, module=task05, name="task05"):
    args = parser_conditional().parse_args(argv)
    string = "{0}_{1}_{2}_{3}".format(name, args.objective, args.stochastic_layers, args.n_samples)      # tasks/task05.py:48
    if args.stochastic_layers == 1:
        return run_training(args, string, lambda **kw: module.CIWAE(200, 100, **kw), args.objective, labelled=True)
    if args.objective == "vae_elbo_kl":
        raise KeyError(args.objective)
    return run_training(args, string, lambda **kw: iwae2.IWAE([200, 100], [100, 50], **kw), args.objective)


if __name__ == "__main__":
    main()
