"""Session / checkpoint / epoch-loop surface of the reference's BaseModel.

Mirrors `model/base.py` of the reference (cited per method).  The TF session is
replaced by `latex_ocr_amd.engine.Engine`; the TF Saver by a single-file archive
written at the reference's path `<out>/model_weights/model.cpkt-<epoch>` (keys =
TF variable names of SURVEY.md Appendix B, Adam slots under `optimize/...`).
"""
import os
import time

import numpy as np

from .utils.general import get_logger, init_dir


def adam_steps_from_powers(beta1_power, beta2_power=None, beta1=0.9, beta2=0.999):
    """Steps taken by a tf.train.AdamOptimizer from its saved accumulators.  TF creates `beta1_power` = beta1 and
    multiplies it by beta1 AFTER every update, so after t steps it holds beta1^(t+1) (same for beta2_power).  In float32
    0.9^t underflows to 0 after ~1000 steps, so a realistically trained checkpoint has beta1_power == 0: fall back to
    beta2_power (0.999^t is representable up to ~87k steps), and past that to a large t whose bias corrections are 1.
    Never raises: a weights restore must not fail on optimizer metadata."""
    def steps(p, beta):
        try:
            p = float(np.asarray(p).reshape(-1)[0])
        except Exception:
            return None
        if not np.isfinite(p) or p <= 0.0 or p >= 1.0:
            return None
        return max(0, int(round(np.log(p) / np.log(beta))) - 1)
    t1 = steps(beta1_power, beta1) if beta1_power is not None else None
    t2 = steps(beta2_power, beta2) if beta2_power is not None else None
    if t1 is not None and t1 < 600:           # float32 0.9^t still carries >= 3 significant digits here
        return t1
    if t2 is not None:
        return t2
    if t1 is not None:
        return t1
    return 1000000


class BaseModel(object):
    def __init__(self, config, dir_output):
        """Reference: model/base.py:12-23."""
        self._config = config
        self._dir_output = dir_output
        init_dir(self._dir_output)
        self.logger = get_logger(self._dir_output + "model.log")
        self.engine = None
        self.startepoch = 0

    def build_train(self, config=None):
        raise NotImplementedError

    def build_pred(self, config=None):
        raise NotImplementedError

    # ------------------------------------------------------------ checkpoints --
    def _dir_model(self):
        return self._dir_output + "model_weights/"

    @staticmethod
    def latest_checkpoint(dir_model):
        """tf.train.latest_checkpoint: path recorded in `<dir>/checkpoint`, else None."""
        index = os.path.join(dir_model, "checkpoint")
        if not os.path.exists(index):
            return None
        with open(index) as f:
            for line in f:
                if line.startswith("model_checkpoint_path:"):
                    name = line.split(":", 1)[1].strip().strip('"')
                    path = name if os.path.isabs(name) else os.path.join(dir_model, name)
                    return path if (os.path.exists(path) or os.path.exists(path + ".index")) else None
        return None

    def init_session(self):
        """Reference: model/base.py:33-48.  Auto-restores the latest checkpoint and sets
        `startepoch` from the text after the first "-" of its path (quirk C-7); if that
        text is not an integer (a "-" in a directory name) the last "-" is used."""
        dir_model = self._dir_model()
        init_dir(dir_model)
        self.ckeck_point = self.latest_checkpoint(dir_model)
        self.startepoch = 0
        if self.ckeck_point is not None:
            self.restore_session(self.ckeck_point)
            try:
                self.startepoch = int(self.ckeck_point[self.ckeck_point.find("-") + 1:])
            except ValueError:
                self.startepoch = int(self.ckeck_point[self.ckeck_point.rfind("-") + 1:])
            self.logger.info("found a checkpoint, resuming at epoch %d" % self.startepoch)

    def restore_session(self, path):
        """Reference: model/base.py:50-59.  `path` may be a checkpoint file or the weights dir."""
        self.logger.info("Reloading the latest trained model...")
        if os.path.isdir(path):
            path = self.latest_checkpoint(path if path.endswith("/") else path + "/")
            if path is None:
                raise IOError("no checkpoint in directory")
        z = self._open_checkpoint(path)
        names = [k for k, _, _ in self.engine.specs]
        missing = [k for k in names if k not in z]
        if missing:
            raise KeyError("checkpoint %s lacks variables: %s" % (path, ", ".join(missing[:4])))
        sd = {"params": {k: np.asarray(z[k], np.float32) for k in names}}
        have_slots = all(("optimize/" + k + "/Adam") in z and ("optimize/" + k + "/Adam_1") in z for k in names)
        if have_slots and ("optimize/adam_t" in z or "optimize/beta1_power" in z):
            off, m, v = 0, np.zeros(self.engine.n_params, np.float32), np.zeros(self.engine.n_params, np.float32)
            for k, shp, _ in self.engine.specs:
                n = int(np.prod(shp))
                m[off:off + n] = np.asarray(z["optimize/" + k + "/Adam"]).reshape(-1)
                v[off:off + n] = np.asarray(z["optimize/" + k + "/Adam_1"]).reshape(-1)
                off += n
            if "optimize/adam_t" in z:
                t = int(np.asarray(z["optimize/adam_t"]).reshape(-1)[0])
            else:
                t = adam_steps_from_powers(z.get("optimize/beta1_power"), z.get("optimize/beta2_power"))
            sd.update(adam_m=m, adam_v=v, adam_t=t)
        self._lr_state = {k[len("lxo/lr_schedule/"):]: z[k] for k in z if k.startswith("lxo/lr_schedule/")}
        self.engine.load_state_dict(sd)

    @staticmethod
    def _open_checkpoint(path):
        """{name: array} of a checkpoint in any of the three containers: this framework's npz file (the default,
        written at the reference's path), safetensors, or a TensorFlow Saver bundle (`<path>.index` +
        `<path>.data-*`, what the reference's model/base.py:61-69 writes) read by latex_ocr_amd/tf_checkpoint.py."""
        if path.endswith(".index"):
            path = path[:-len(".index")]
        if os.path.exists(path + ".index") and not os.path.isfile(path):
            from ..tf_checkpoint import read_bundle
            return read_bundle(path)
        if path.endswith(".safetensors"):
            from safetensors.numpy import load_file
            return load_file(path)
        with np.load(path, allow_pickle=False) as z:
            return {k: z[k] for k in z.files}

    def save_session(self, epoch):
        """Reference: model/base.py:61-69 (Saver(max_to_keep=1): older checkpoints are removed;
        Adam slots are saved, LR-schedule state and best_score are not)."""
        if getattr(self, "dist", None) is not None and self.dist.rank != 0:
            return                                    # data parallel: parameters are replicated, rank 0 writes
        dir_model = self._dir_model()
        init_dir(dir_model)
        self.logger.info("- Saving model...")
        sd = self.engine.state_dict()
        arrays = dict(sd["params"])
        off = 0
        for k, shp, _ in self.engine.specs:
            n = int(np.prod(shp))
            arrays["optimize/" + k + "/Adam"] = sd["adam_m"][off:off + n].reshape(shp)
            arrays["optimize/" + k + "/Adam_1"] = sd["adam_v"][off:off + n].reshape(shp)
            off += n
        arrays["optimize/adam_t"] = np.int64(sd["adam_t"])
        sched = getattr(self, "_lr_schedule", None)
        if sched is not None:       # extension: the reference loses the schedule on resume
            for k, v in sched.state_dict().items():
                arrays["lxo/lr_schedule/" + k] = np.float64(v)
        name = "model.cpkt-%d" % epoch
        with open(dir_model + name, "wb") as f:
            np.savez(f, **arrays)
        for old in os.listdir(dir_model):
            if old.startswith("model.cpkt-") and old != name:
                os.remove(dir_model + old)
        with open(dir_model + "checkpoint", "w") as f:
            f.write('model_checkpoint_path: "%s"\n' % name)
        self.logger.info("- Saved model in {}".format(dir_model))

    # ------------------------------------------------------------- epoch loop --
    def train(self, config, train_set, val_set, lr_schedule):
        """Reference: model/base.py:95-138.  Epochs below `startepoch` are skipped; the model is
        saved when the epoch score is >= the best so far; early stop on lr_schedule."""
        best_score = None
        self._lr_schedule = lr_schedule
        if getattr(self, "_lr_state", None) and getattr(config, "resume_lr_schedule", True):
            lr_schedule.load_state_dict(self._lr_state)
            self.logger.info("- restored the learning-rate schedule (lr {:.6g})".format(lr_schedule.lr))
        for epoch in range(config.n_epochs):
            if epoch < self.startepoch:
                continue
            tic = time.time()
            self.logger.info("Epoch {:}/{:}".format(epoch + 1, config.n_epochs))
            score = self._run_train(config, train_set, val_set, epoch, lr_schedule)
            if best_score is None or score >= best_score:
                best_score = score
                self.logger.info("- New best score ({:04.2f})!".format(best_score))
                self.save_session(epoch)
            if lr_schedule.stop_training:
                self.logger.info("- Early Stopping.")
                break
            self.logger.info("- Elapsed time: {:04.2f}, lr: {:04.5f}".format(time.time() - tic, lr_schedule.lr))
        return best_score

    def evaluate(self, config, test_set):
        """Reference: model/base.py:158-191."""
        scores = self._run_evaluate(config, test_set)
        msg = " - ".join(["{} {:04.2f}".format(k, v) for k, v in scores.items()])
        self.logger.info("- Eval: {}".format(msg))
        return scores
