"""Training driver with the reference's CLI (main.py:16-25 of nbip/IWAE): same flags, same
schedule, same evaluation protocol, on the MI355X-native step.

    python main.py --stochastic_layers 1 --n_samples 50 --batch_size 20 --objective iwae_elbo

Differences that are forced by the environment, not by design: MNIST is read from a local
mnist.npz (IWAE_MNIST_PATH, ~/.keras/datasets/mnist.npz) because the Keras download is not
available offline, with a synthetic grey-level stand-in otherwise; scalars go to a CSV under
/tmp/iwae/<run>/ instead of TensorBoard; --gpu selects the HIP device ordinal.

Data-parallel (BASELINE configs[4]; the reference is single-device, main.py:32): launch one process per GPU with torchrun,

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29500 \
           main.py --n_samples 50 --batch_size 8192

--batch_size is then the GLOBAL batch: rank r trains on images [r*B/N, (r+1)*B/N) of every batch (same shuffle on every
rank, noise keyed by the global image index), the library all-reduces the gradient with RCCL (iwae_comm_init) and every
rank applies the same Adam step.  A gloo group (CPU) only ships the RCCL ids and sums the test-set estimate; it is created
before anything touches the GPU.
"""
import argparse
import csv
import datetime
import os
import time

import numpy as np

from iwae_amd import iwae1, iwae2, utils
from iwae_amd.optimizers import Adam

parser = argparse.ArgumentParser()
parser.add_argument("--stochastic_layers", type=int, default=1, choices=[1, 2], help="number of stochastic layers in the model")
parser.add_argument("--n_samples", type=int, default=5, help="number of importance samples")
parser.add_argument("--batch_size", type=int, default=20, help="batch size")
parser.add_argument("--epochs", type=int, default=-1,
                    help="numper of epochs, if set to -1 number of epochs "
                         "will be set based on the learning rate scheme from the paper")
parser.add_argument("--objective", type=str, default="iwae_elbo", choices=["vae_elbo", "iwae_elbo", "iwae_eq14", "vae_elbo_kl"])
parser.add_argument("--gpu", type=str, default='0', help="Choose GPU")


def _init_data_parallel():
    """(rank, world, dist or None) from torchrun's environment; world 1 without it."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world <= 1:
        return 0, 1, None
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=world)      # CPU only: ships ids, sums scalars
    return int(os.environ["RANK"]), world, dist


def main(argv=None):
    args = parser.parse_args(argv)
    string = "main_{0}_{1}_{2}".format(args.objective, args.stochastic_layers, args.n_samples)

    def make_model(**kw):
        if args.stochastic_layers == 1:
            return iwae1.IWAE(200, 100, **kw)
        if args.objective == "vae_elbo_kl":
            raise KeyError(args.objective)      # src/iwae2.py:154-173 has no such key
        return iwae2.IWAE([200, 100], [100, 50], **kw)

    return run_training(args, string, make_model, args.objective)


def run_training(args, string, make_model, objective, labelled=False, report=None):
    """The training protocol of the reference's drivers (main.py:38-184; tasks/task02.py:110-259, task04.py / task05.py:200-345 are the same loop
    around another model class): seeds, the 8-stage learning-rate schedule, per-epoch dynamic binarisation + shuffle, a test evaluation every 200
    steps, final weights, the k = 5000 test-set estimate.  make_model(**kw) builds the model (kw: device, output_bias, world_size, rank);
    labelled: the model's steps take (x, y) (the conditional models); report: the scalar printed as "ELBO" (default: the objective)."""
    rank, world, dist = _init_data_parallel()
    if rank == 0:
        print(args)
    device = int(os.environ["LOCAL_RANK"]) if world > 1 else int(str(args.gpu).split(",")[0])
    report = report or objective

    # ---- set random seeds (main.py:40-41)
    np.random.seed(123)

    # ---- number of passes over the data, see bottom of page 6 in [1] (main.py:44-56)
    if args.epochs == -1:
        epochs = 0
        learning_rate_dict = {}
        for i in range(8):
            learning_rate_dict[epochs] = 0.001 * 10 ** (-i / 7)
            epochs += 3 ** i
    else:
        epochs = args.epochs
        learning_rate_dict = {0: 0.0001}

    # ---- load data (main.py:59-65)
    data = utils.load_mnist()
    if data is not None:
        (Xtrain, ytrain), (Xtest, ytest) = data
        Ntrain, Ntest = Xtrain.shape[0], Xtest.shape[0]
        Xtrain = Xtrain.reshape(Ntrain, -1) / 255
        Xtest = Xtest.reshape(Ntest, -1) / 255
    else:
        print("NOTE: no local mnist.npz found (set IWAE_MNIST_PATH); using synthetic MNIST-like data")
        Xtrain, Xtest = utils.synthetic_mnist()
        Ntrain, Ntest = Xtrain.shape[0], Xtest.shape[0]
        lab = np.random.default_rng(123)
        ytrain, ytest = lab.integers(0, 10, Ntrain), lab.integers(0, 10, Ntest)      # (stand-in classes for the conditional models)

    n_samples = args.n_samples
    batch_size = args.batch_size
    if batch_size % world:
        raise SystemExit("--batch_size {0} (the global batch) must be divisible by the {1} ranks".format(batch_size, world))
    steps_pr_epoch = Ntrain // batch_size
    total_steps = steps_pr_epoch * epochs

    current_time = datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
    log_dir = "/tmp/iwae/{0}/".format(string) + current_time
    os.makedirs(log_dir, exist_ok=True)
    log_f = open(os.path.join(log_dir, "scalars.csv") if rank == 0 else os.devnull, "w", newline="")
    log_w = None

    # ---- instantiate the model, optimizer and metrics (main.py:84-94)
    model = make_model(device=device, output_bias=utils.get_bias(Xtrain), world_size=world, rank=rank)

    if world > 1:      # the gradient exchange happens inside the library from here on (RCCL over xGMI)
        from iwae_amd.parallel import init_in_library_exchange
        why = init_in_library_exchange(model._net, rank, world)      # on every rank or on none (collective agreement)
        if why is not None:
            raise RuntimeError("data-parallel training needs the library's RCCL exchange on every rank: " + why)

    optimizer = Adam(learning_rate_dict[0], epsilon=1e-4)
    if rank == 0:
        print("Initial learning rate: ", optimizer.learning_rate.numpy())

    # ---- binarize the test data once (main.py:108)
    Xtest = utils.bernoullisample(Xtest)

    # ---- the training set stays in HBM; shuffling order comes from the host RNG, binarisation and the
    #      batch gather run on the device (the reference re-binarises 47M pixels on the host every epoch)
    if labelled:
        model.set_dataset(Xtrain, ytrain)      # (x, y) pairs: tasks/task05.py:296-322
    else:
        model.set_dataset(Xtrain)

    start = time.time()
    for epoch in range(epochs):
        # ---- binarize the training data at the start of each epoch + shuffle (main.py:117-120)
        model.begin_epoch(epoch, np.random.permutation(Ntrain))

        if args.epochs == -1 and epoch in learning_rate_dict:
            new_learning_rate = learning_rate_dict[epoch]
            old_learning_rate = optimizer.learning_rate.numpy()
            if rank == 0:
                print("Changing learning rate from {0} to {1}".format(old_learning_rate, new_learning_rate))
            optimizer.learning_rate.assign(new_learning_rate)

        for _step, lo in enumerate(range(0, Ntrain, batch_size)):
            step = _step + steps_pr_epoch * epoch
            beta = 1.0
            nb = min(batch_size, Ntrain - lo)
            if world > 1:
                nb = (nb // world)                       # this rank's shard of the global batch (a ragged last batch drops < world images)
                model._net.set_step(step, rank * nb)     # noise keyed by the global image index: N ranks draw what one rank would
                if nb == 0:
                    continue
                res = model.train_step_dataset(lo + rank * nb, nb, n_samples, beta, optimizer, objective=objective)
            else:
                res = model.train_step_dataset(lo, nb, n_samples, beta, optimizer, objective=objective)

            if step % 200 == 0 and rank == 0:
                test_res = model.val_step(Xtest, ytest, n_samples, beta) if labelled else model.val_step(Xtest, n_samples, beta)
                row = {"split": "train", **model.write_to_tensorboard(res, step)}
                row_t = {"split": "test", **model.write_to_tensorboard(test_res, step)}
                if log_w is None:
                    log_w = csv.DictWriter(log_f, fieldnames=list(row_t.keys()), restval="", extrasaction="ignore")
                    log_w.writeheader()
                log_w.writerow(row)
                log_w.writerow(row_t)
                log_f.flush()
                took = time.time() - start
                start = time.time()
                print("epoch {0}/{1}, step {2}/{3}, train ELBO: {4:.2f}, val ELBO: {5:.2f}, time: {6:.2f}"
                      .format(epoch, epochs, step, total_steps, res[report].numpy(), test_res[report], took))

    # ---- save final weights (main.py:165)
    if rank == 0:
        model.save_weights('/tmp/iwae/{0}/final_weights'.format(string))

    # ---- test-set llh estimate using 5000 samples (main.py:170-184); data-parallel: every rank takes a slice of the test set
    L = 5000
    llh_of = (lambda X, y: model.eval_llh(X, y, L)) if labelled else (lambda X, y: model.eval_llh(X, L))
    if world > 1:
        import torch
        mine, ymine = Xtest[rank::world], ytest[rank::world] if labelled else None
        model._net.set_step(1 << 20, 0)
        part = torch.tensor([llh_of(mine, ymine) * mine.shape[0], float(mine.shape[0])], dtype=torch.float64)
        dist.all_reduce(part)
        test_set_llh = float(part[0] / part[1])
        dist.destroy_process_group()
    else:
        test_set_llh = llh_of(Xtest, ytest if labelled else None)
    if rank == 0:
        print("Test-set {0} sample log likelihood estimate: {1:.4f}".format(L, test_set_llh))
    return test_set_llh


if __name__ == "__main__":
    main()
