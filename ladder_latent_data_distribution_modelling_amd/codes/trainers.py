"""Epoch drivers with the reference's class names (codes/trainers.py:12-83, 130-209)."""
import numpy as np

from .base import BaseTrain, BaseTrain_joint  # noqa: F401
from .data_loader import BatchIterator, DeviceBatchIterator


class _JointEpochMixin:
    """The per-epoch sequence shared by both trainers (trainers.py:23-83 and 141-198); the three places where the reference's two
    trainers differ are class attributes / hooks overridden by CelebATrainer_joint_training."""
    EPOCH_BANNER = "{}/{}:"                                                   # trainers.py:25
    AVERAGE_MSG = "Average overall negative ELBO loss:\ntrain: {:.4f}, val: {:.4f}"   # trainers.py:66-68

    def _val_gates(self):
        """(run val_step('VAE'), run val_step('prior')) of the validation loop: MNIST trainer, trainers.py:57-63."""
        cfg = self.config
        return True, self.cur_epoch > int(cfg["sg_pretraining"]) - 1 and cfg["prior"] in ("ours", "hierarchical", "vampPrior")

    def _iterators(self):
        raise NotImplementedError

    def _mid_epoch(self, i):
        """Hook behind iteration i of the hot loop (the CelebA trainer records its mid-epoch test steps here)."""

    def _make_iterator(self, key, images, shuffle=True):
        """config["device_resident_data"] (default 1): the rank's shard of the split is uploaded to HBM once (uint8 for CelebA)
        and minibatches are gathered + normalised on the device; 0 keeps the host iterator (float32 numpy batches)."""
        bs = int(self.config["batch_size"])
        if not int(self.config.get("device_resident_data", 1)):
            img = images() if callable(images) else images
            if img.dtype == np.uint8:
                img = img.astype(np.float32) * np.float32(1.0 / 255)
            return BatchIterator(img, bs, seed=self.cur_epoch, shuffle=shuffle)
        cache = self.__dict__.setdefault("_dev_sets", {})
        if key not in cache:
            cache[key] = DeviceBatchIterator(images() if callable(images) else images, bs, device=self.engine.ctx.device).data
        return DeviceBatchIterator(cache[key], bs, seed=self.cur_epoch, shuffle=shuffle, device=self.engine.ctx.device)

    def _prior_training_on(self):
        cfg = self.config
        return (self.cur_epoch > int(cfg["sg_pretraining"]) - 1 and cfg["prior"] in ("ours", "hierarchical", "vampPrior")
                and int(cfg["TRAIN_prior"]) == 1)

    def train_epoch(self):
        cfg = self.config
        self.cur_epoch += 1
        print(self.EPOCH_BANNER.format(self.cur_epoch, cfg["num_epochs"]))
        train_it, val_it = self._iterators()
        self.compute_cur_lr()
        train_loss_cur_epoch = 0.0
        for i in range(self.n_train_iter):                                  # HOT LOOP (trainers.py:33-40, 148-155)
            batch = train_it.next()
            if int(cfg["TRAIN_VAE"]) == 1:
                loss = self.train_step_ae(cur_lr=self.cur_lr, batch_data=batch)
                self.train_loss.append(loss)
                train_loss_cur_epoch += loss
            if self._prior_training_on():
                self.train_step_prior(batch_data=batch)
            self._mid_epoch(i)
        self.flush()                                                        # record lists of the last iteration (async_fetch)
        if int(cfg["TRAIN_VAE"]) == 1:
            self.train_loss_ave_epoch.append(train_loss_cur_epoch / max(self.n_train_iter, 1))
            self.iter_epochs_list.append(len(self.train_loss) - 1)
        if self.cur_epoch > int(cfg["sg_pretraining"]) - 1 and cfg["prior"] in ("ours", "GMM"):
            self.fit_GM(iterator=train_it)
        self.generate_samples_from_prior()
        self.test_step(batch_data=self.test_batch, print_result=True)
        val_loss_cur_epoch = 0.0
        val_vae, val_prior = self._val_gates()
        for i in range(self.n_val_iter):
            vb = val_it.next()
            if val_vae:
                val_loss_cur_epoch += self.val_step(batch_data=vb, model_to_train="VAE")
            if val_prior:
                self.val_step(batch_data=vb, model_to_train="prior")
        self.val_loss_ave_epoch.append(val_loss_cur_epoch / max(self.n_val_iter, 1))
        if int(cfg["TRAIN_VAE"]) == 1:
            print(self.AVERAGE_MSG.format(self.train_loss_ave_epoch[self.cur_epoch - 1], self.val_loss_ave_epoch[self.cur_epoch - 1]))
        self.save_variables_VAE()


class MNISTTrainer_joint_training(_JointEpochMixin, BaseTrain_joint):
    def __init__(self, sess, model, data, config):
        super().__init__(sess, model, data, config)
        self.test_batch = self.data.test_set["image"]
        world = self.engine.ctx.comm.world
        self.n_train_iter = self.data.n_train // (int(config["batch_size"]) * world)
        self.n_val_iter = self.data.n_val // (int(config["batch_size"]) * world)

    def _shard(self, images):
        c = self.engine.ctx.comm
        return images[c.rank::c.world] if c.on else images

    def _iterators(self):
        return (self._make_iterator("train", lambda: self._shard(self.data.train_set["image"])),
                self._make_iterator("val", lambda: self._shard(self.data.val_set["image"])))

    def compute_cur_lr(self):
        self.cur_lr = float(self.config["learning_rate_ae"]) * (0.99 ** (self.cur_epoch - 1))     # trainers.py:30


class CelebATrainer_joint_training(_JointEpochMixin, BaseTrain_joint):
    EPOCH_BANNER = "Training epoch: {}/{}"                                    # trainers.py:143
    AVERAGE_MSG = "Average:\ntrain: {:.4f}, val: {:.4f}"                      # trainers.py:186-188

    def _val_gates(self):
        """trainers.py:178-183: the VAE validation pass only when the VAE trains, the prior pass only when the prior trains."""
        cfg = self.config
        return (int(cfg["TRAIN_VAE"]) == 1,
                self.cur_epoch > int(cfg["sg_pretraining"]) - 1 and int(cfg["TRAIN_prior"]) == 1
                and cfg["prior"] in ("ours", "hierarchical", "vampPrior"))

    def __init__(self, sess, model, data, config):
        super().__init__(sess, model, data, config)
        bs = int(config["batch_size"])
        self.test_batch = self.data.celeba_images("test", limit=bs)[:bs]
        c = self.engine.ctx.comm
        # this rank's shard only (records rank::world), uint8 as on disk
        self._train = self.data.celeba_images_u8("train", rank=c.rank, world=c.world)
        self._val = self.data.celeba_images_u8("val", rank=c.rank, world=c.world)
        n_train = self.data.n_train if not self.data.synthetic else self._train.shape[0] * c.world
        n_val = self.data.n_val if not self.data.synthetic else self._val.shape[0] * c.world
        self.n_train_iter = min(n_train // (bs * c.world), self._train.shape[0] // bs)
        self.n_val_iter = min(n_val // (bs * c.world), self._val.shape[0] // bs)
        # trainers.py:139: iterations behind which the reference evaluates (and plots) the test batch inside the epoch
        step = max(1, self.n_train_iter // max(1, int(config.get("num_iter_to_plot", 1))))
        self.idx_check_point = np.arange(0, self.n_train_iter - 1, step)

    def _mid_epoch(self, i):
        """trainers.py:156-158: test_step(test_batch) at the check-point iterations -- the figure is out of scope, the state it records
        (`test_sigma` -> result-npz `sigma`, `output_test`) is not: one entry per check point, as in a reference run."""
        if int(self.config.get("num_iter_to_plot", 1)) > 1 and np.any(self.idx_check_point == i):
            self.test_step(batch_data=self.test_batch, print_result=False)

    def _iterators(self):
        return (self._make_iterator("train", lambda: self._train),
                # the reference re-initialises the SAME (shuffling) iterator on the validation file (trainers.py:173-174, models.py:354-386)
                self._make_iterator("val", lambda: self._val, shuffle=True))

    def compute_cur_lr(self):
        """Piecewise schedule of codes/trainers.py:200-209."""
        e, lr0 = self.cur_epoch, float(self.config["learning_rate_ae"])
        if e <= 25:
            self.cur_lr = lr0 * (0.99 ** (e - 1))
        elif e <= 50:
            self.cur_lr = lr0 / 2 * (0.99 ** (e - 25))
        elif e <= 75:
            self.cur_lr = lr0 / 5 * (0.99 ** (e - 50))
        else:
            self.cur_lr = lr0 / 10 * (0.99 ** (e - 75))
