"""Training driver with the reference's CLI (main.py:16-25 of nbip/IWAE): same flags, same
schedule, same evaluation protocol, on the MI355X-native step.

    python main.py --stochastic_layers 1 --n_samples 50 --batch_size 20 --objective iwae_elbo

Differences that are forced by the environment, not by design: MNIST is read from a local
mnist.npz (IWAE_MNIST_PATH, ~/.keras/datasets/mnist.npz) because the Keras download is not
available offline, with a synthetic grey-level stand-in otherwise; scalars go to a CSV under
/tmp/iwae/<run>/ instead of TensorBoard; --gpu selects the HIP device ordinal.
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


def main(argv=None):
    args = parser.parse_args(argv)
    print(args)
    string = "main_{0}_{1}_{2}".format(args.objective, args.stochastic_layers, args.n_samples)
    device = int(str(args.gpu).split(",")[0])

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

    objective = args.objective
    n_samples = args.n_samples
    batch_size = args.batch_size
    steps_pr_epoch = Ntrain // batch_size
    total_steps = steps_pr_epoch * epochs

    current_time = datetime.datetime.now().strftime("%Y%m%d-%H%M%S")
    log_dir = "/tmp/iwae/{0}/".format(string) + current_time
    os.makedirs(log_dir, exist_ok=True)
    log_f = open(os.path.join(log_dir, "scalars.csv"), "w", newline="")
    log_w = None

    # ---- instantiate the model, optimizer and metrics (main.py:84-94)
    bias = utils.get_bias(Xtrain)
    if args.stochastic_layers == 1:
        model = iwae1.IWAE(200, 100, device=device, output_bias=bias)
    else:
        if objective == "vae_elbo_kl":
            raise KeyError(objective)      # src/iwae2.py:154-173 has no such key
        model = iwae2.IWAE([200, 100], [100, 50], device=device, output_bias=bias)

    optimizer = Adam(learning_rate_dict[0], epsilon=1e-4)
    print("Initial learning rate: ", optimizer.learning_rate.numpy())

    # ---- binarize the test data once (main.py:108)
    Xtest = utils.bernoullisample(Xtest)

    # ---- the training set stays in HBM; shuffling order comes from the host RNG, binarisation and the
    #      batch gather run on the device (the reference re-binarises 47M pixels on the host every epoch)
    model.set_dataset(Xtrain)

    start = time.time()
    for epoch in range(epochs):
        # ---- binarize the training data at the start of each epoch + shuffle (main.py:117-120)
        model.begin_epoch(epoch, np.random.permutation(Ntrain))

        if args.epochs == -1 and epoch in learning_rate_dict:
            new_learning_rate = learning_rate_dict[epoch]
            old_learning_rate = optimizer.learning_rate.numpy()
            print("Changing learning rate from {0} to {1}".format(old_learning_rate, new_learning_rate))
            optimizer.learning_rate.assign(new_learning_rate)

        for _step, lo in enumerate(range(0, Ntrain, batch_size)):
            step = _step + steps_pr_epoch * epoch
            beta = 1.0
            res = model.train_step_dataset(lo, min(batch_size, Ntrain - lo), n_samples, beta, optimizer, objective=objective)

            if step % 200 == 0:
                test_res = model.val_step(Xtest, n_samples, beta)
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
                      .format(epoch, epochs, step, total_steps, res[objective].numpy(), test_res[objective], took))

    # ---- save final weights (main.py:165)
    model.save_weights('/tmp/iwae/{0}/final_weights'.format(string))

    # ---- test-set llh estimate using 5000 samples (main.py:170-184)
    L = 5000
    test_set_llh = model.eval_llh(Xtest, L)
    print("Test-set {0} sample log likelihood estimate: {1:.4f}".format(L, test_set_llh))
    return test_set_llh


if __name__ == "__main__":
    main()
