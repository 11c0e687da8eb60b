"""Img2SeqModel: the reference's model facade over the MI355X-native engine.

Mirrors `model/img2seq.py` of the reference method by method (cited); the single
`sess.run` boundary (:169, :236, :263) becomes calls into liblxo.so through
`latex_ocr_amd.engine.Engine`.  Inputs are the reference's: lists of variable-shape
uint8 [H,W,1] arrays and lists of int lists; the facade owns padding.
"""
import numpy as np

from .base import BaseModel
from .evaluation.text import score_files, truncate_end, write_answers
from .params import dims_from_config
from .utils.general import Config, Progbar, minibatches
from .utils.image import pad_batch_images
from .utils.text import pad_batch_formulas


class Img2SeqModel(BaseModel):
    def __init__(self, config, dir_output, vocab):
        """Reference: model/img2seq.py:23-32."""
        super(Img2SeqModel, self).__init__(config, dir_output)
        self._vocab = vocab
        self.dist = None            # latex_ocr_amd.dist.DataParallel when launched one process per GPU (attach_dist)

    def attach_dist(self, dist):
        """Data-parallel training: every rank builds the same model, trains on its share of each
        shape bucket and all-reduces gradients; evaluation / checkpoints are written by rank 0 only."""
        self.dist = dist

    # ------------------------------------------------------------------ build --
    def _build_engine(self):
        from ..engine import Engine
        cfg = self._config
        self.engine = Engine(self._vocab.n_tok, dims=dims_from_config(cfg),
                             dtype=getattr(cfg, "compute_dtype", "bf16"),
                             device=getattr(cfg, "device", "cuda:0") if str(getattr(cfg, "device", "")).startswith("cuda") else "cuda:0",
                             seed=getattr(cfg, "seed", 0),
                             beam=getattr(cfg, "beam_size", 1) if getattr(cfg, "decoding", "greedy") == "beam_search" else 1,
                             max_steps=getattr(cfg, "max_length_formula", 150) + 2)

    def build_train(self, config):
        """Reference: model/img2seq.py:34-40."""
        self.logger.info("Building model...")
        self._build_engine()
        self._lr_method = config.lr_method.lower()
        self.engine.set_optimizer(self._lr_method)               # img2seq.py:95-109
        self._clip = getattr(config, "clip", -1)
        self.init_session()
        self.logger.info("- done.")

    def build_pred(self):
        """Reference: model/img2seq.py:42-46."""
        self.logger.info("Building model...")
        self._build_engine()
        self.init_session()
        self.logger.info("- done.")

    # ------------------------------------------------------------------ feeds --
    def _get_feed_dict(self, img, formula=None, lr=None, dropout=1):
        """Reference: model/img2seq.py:125-142.  `dropout` is a KEEP probability (quirk C-5);
        every value >= 1 is the identity (the shipped configs use 1 and 127); values in (0, 1) drop h and o
        in the decoder cell (attention_cell.py:72,83) -- the encoder receives the placeholder but never uses it."""
        if not dropout > 0:
            raise ValueError("dropout is a keep probability and must be > 0, got {}".format(dropout))
        fd = {"img": pad_batch_images(img), "dropout": dropout}
        if formula is not None:
            f, l = pad_batch_formulas(formula, self._vocab.id_pad, self._vocab.id_end)
            fd["formula"], fd["formula_length"] = f, l
        if lr is not None:
            fd["lr"] = lr
        return fd

    # ------------------------------------------------------------------ train --
    def _run_train(self, config, train_set, val_set, epoch, lr_schedule):
        """Reference: model/img2seq.py:144-196.  The feed runs on the background input pipeline (pipeline.py:
        pinned staging + copy stream, `prefetch_depth` batches ahead; config.prefetch_depth = 0 restores the
        reference's synchronous feed); under data parallelism every rank takes its slice of each shape bucket."""
        from ..pipeline import Prefetcher, ShardedBuckets
        batch_size = config.batch_size
        world = self.dist.world if self.dist is not None else 1
        rank = self.dist.rank if self.dist is not None else 0
        nbatches = (len(train_set) + batch_size * world - 1) // (batch_size * world)
        keep = getattr(config, "dropout", 1)
        if not keep > 0:
            raise ValueError("dropout is a keep probability and must be > 0, got {}".format(keep))
        depth = int(getattr(config, "prefetch_depth", 2))
        # steps per pass are counted once per (dataset, batch size, world): another train() call with other arguments must not reuse the count
        key = (id(train_set), batch_size, world)
        cached = getattr(self, "_dp_nbatches", None)
        batches = ShardedBuckets(train_set, batch_size, world, rank, n_steps=cached[1] if cached and cached[0] == key else None) if world > 1 else None
        if batches is not None:
            nbatches = len(batches)        # steps this pass really takes (per shape bucket), what the LR schedule was scaled with
            self._dp_nbatches = (key, nbatches)   # the same set every epoch: counted once
        prog = Progbar(nbatches)
        if depth > 0:
            feed = Prefetcher(train_set, batch_size, self._vocab.id_pad, self._vocab.id_end, device=self.engine.device,
                              depth=depth, batches=batches)
            it = ((b.img, b.formula, b.lengths) for b in feed)
        else:
            def sync_feed():
                for img, formula in (batches if batches is not None else minibatches(train_set, batch_size)):
                    fd = self._get_feed_dict(img, formula=formula)
                    yield fd["img"], fd["formula"], fd["formula_length"]
            it = sync_feed()
        for i, (img, formula, lengths) in enumerate(it):
            loss_eval = self.engine.train_step(img, formula, lengths, lr_schedule.lr, clip=self._clip, dist=self.dist,
                                               dropout=keep)
            prog.update(i + 1, [("loss", loss_eval), ("perplexity", np.exp(loss_eval)), ("lr", lr_schedule.lr)])
            lr_schedule.update(batch_no=epoch * nbatches + i)
        self.logger.info("- Training: {}".format(prog.info))
        sub = "formulas_val/" if rank == 0 else "formulas_val_rank%d/" % rank    # every rank scores (keeps LR schedules equal)
        config_eval = Config({"dir_answers": self._dir_output + sub, "batch_size": config.batch_size})
        scores = self.evaluate(config_eval, val_set)
        score = scores["perplexity"]
        if self.dist is not None:
            # every rank scored its own copy of the validation set with atomically (order-dependently) accumulated statistics:
            # rank 0's number decides, so that score-driven decay / early stopping cannot diverge between ranks
            score = self.dist.broadcast_scalar(score)
        lr_schedule.update(score=score)
        return score

    def _run_evaluate(self, config, test_set):
        """Reference: model/img2seq.py:198-213."""
        files, perp = self.write_prediction(config, test_set)
        scores = score_files(files[0], files[1])
        scores["perplexity"] = perp
        return scores

    def _decode(self, img):
        """pred_test.ids of the decode graph (decoder.py:60-70), shaped [B, k, T'] as after
        img2seq.py:238-241."""
        cfg = self._config
        max_iter = getattr(cfg, "max_length_formula", 150) + 1          # decoder.py:70
        if getattr(cfg, "decoding", "greedy") == "beam_search":
            self._div_calls = getattr(self, "_div_calls", 0) + 1
            ids, par = self.engine.beam_decode(img, self._vocab.id_end, cfg.beam_size, max_iter=max_iter, return_parents=True,
                                               div_gamma=getattr(cfg, "div_gamma", 1), div_prob=getattr(cfg, "div_prob", 0),
                                               div_seed=self._div_calls)         # decoder.py:67-68
            if getattr(cfg, "beam_backtrace", False):      # extension: follow parents (the reference never does, quirk C-1)
                from .utils.text import beam_backtrace
                ids = beam_backtrace(ids, par)
            return np.transpose(ids, [0, 2, 1])
        ids = self.engine.greedy_decode(img, self._vocab.id_end, max_iter=max_iter)
        return np.expand_dims(ids, axis=1)

    def write_prediction(self, config, test_set):
        """Reference: model/img2seq.py:215-254.  perplexity is NEGATED (quirk C-2)."""
        k = self._config.beam_size if getattr(self._config, "decoding", "greedy") == "beam_search" else 1
        refs, hyps = [], [[] for _ in range(k)]
        n_words, ce_words = 0, 0.0
        for img, formula in minibatches(test_set, config.batch_size):
            fd = self._get_feed_dict(img, formula=formula, dropout=1)
            ce, n = self.engine.evaluate_batch(fd["img"], fd["formula"], fd["formula_length"])
            ids_eval = self._decode(fd["img"])
            n_words += n
            ce_words += ce
            for form, preds in zip(formula, ids_eval):
                refs.append(form)
                for i, pred in enumerate(preds):
                    hyps[i].append(pred)
        files = write_answers(refs, hyps, self._vocab.id_to_tok, config.dir_answers, self._vocab.id_end)
        perp = -np.exp(ce_words / float(n_words))
        return files, perp

    def predict_batch(self, images):
        """Reference: model/img2seq.py:256-276."""
        fd = self._get_feed_dict(images, dropout=1)
        ids_eval = self._decode(fd["img"])
        hyps = [[] for _ in range(ids_eval.shape[1])]
        for preds in ids_eval:
            for i, pred in enumerate(preds):
                p = truncate_end(pred, self._vocab.id_end)
                hyps[i].append(" ".join(self._vocab.id_to_tok[int(idx)] for idx in p))
        return hyps

    def predict(self, img):
        """Reference: model/img2seq.py:278-285."""
        return [hyp[0] for hyp in self.predict_batch([img])]

    def predict_with_attention(self, img):
        """Best hypothesis of one image plus its per-step attention maps [T', H', W'] -- the data the reference gathers in the global
        `ctx_vector` through tf.py_func (attention_mechanism.py:96-121) for visualize_attention.py -- under WHATEVER config.decoding says
        (the shipped configs/model.json:13-14 decodes with beam_search, k = 2).
        Beam search: the reference's visualiser draws row 0 of the merged batch x beam tensor of every step (`attentionVector[0]`,
        visualize_attention.py:55), for every step the loop ran -- it ends when ALL beams have finished, so there are more slices than tokens
        in the best hypothesis (51 slices for 48 tokens at k = 2); that is what comes back here.  With the
        `beam_backtrace` extension each step's map is the one of the row its best token was read off (parents[t][0])."""
        fd = self._get_feed_dict([img], dropout=1)
        cfg = self._config
        max_iter = getattr(cfg, "max_length_formula", 150) + 1
        if getattr(cfg, "decoding", "greedy") == "beam_search":
            self._div_calls = getattr(self, "_div_calls", 0) + 1
            ids, par, alpha = self.engine.beam_decode(fd["img"], self._vocab.id_end, cfg.beam_size, max_iter=max_iter,
                                                      div_gamma=getattr(cfg, "div_gamma", 1), div_prob=getattr(cfg, "div_prob", 0),
                                                      div_seed=self._div_calls, return_attention=True)
            if getattr(cfg, "beam_backtrace", False):
                from .utils.text import beam_backtrace
                maps = np.stack([alpha[0, t, par[0, t, 0]] for t in range(alpha.shape[1])])
                ids = beam_backtrace(ids, par)
            else:
                maps = alpha[0, :, 0]
            best = ids[0, :, 0]
        else:
            ids, alpha = self.engine.greedy_decode(fd["img"], self._vocab.id_end, max_iter=max_iter, return_attention=True)
            best, maps = ids[0], alpha[0]
        p = truncate_end(best, self._vocab.id_end)
        return " ".join(self._vocab.id_to_tok[int(i)] for i in p), maps
