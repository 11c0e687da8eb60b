"""The `getModel()` hook of the reference's PyTorch scaffold, filled in with this repository's engine.

Reference: `/root/reference/model/base_torch.py:110-117` declares `getModel(model_name)` ("return your model that
inherits from torch.nn") and `:74-78` moves it to the device; `/root/reference/model/img2seq_torch.py:64-83` is the
(unfinished) Img2Seq subclass, `:129-163` its `getLoss`.  The reference never completed that path: its `Img2Seq`,
`EncoderCNN(config)` and `DecoderWithAttention` do not fit together.  What a maintainer of that scaffold needs is an
`nn.Module` whose parameters `torch.optim` can update and whose forward / backward run the model -- here the MI355X
hot path behind the C ABI:

    model = Img2SeqModel(config, dir_output, vocab).getModel("Img2Seq")      # nn.Module
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)                       # or any torch optimizer
    loss = model(img_u8, formula_ids, lengths)                                # scalar, mean CE over unmasked tokens
    opt.zero_grad(); loss.backward(); opt.step()

* the module owns ONE flat f32 `nn.Parameter` that IS the engine's parameter buffer (TF variable order, `params.py`);
  `named_views()` exposes the per-variable views under the TF names for inspection / checkpointing;
* `Img2SeqFunction` (a `torch.autograd.Function`) runs `lxo_encoder_fwd` + `lxo_decoder_train_fwd` + `lxo_ce_loss_fwd_bwd`
  in `forward` and `lxo_decoder_train_bwd` + `lxo_encoder_bwd` in `backward`, returning the engine's gradient buffer
  (scaled by the incoming gradient) as the gradient of the flat parameter;
* after an optimizer has written the parameter in place, the next forward refreshes the packed GEMM operands
  (`lxo_pack_weights`); the tensor's version counter tells.

There is no CPU fallback: the module needs the HIP library (or, in the CPU tests, the SIMT interpreter build of the same
sources passed as `lib=`).
"""
import numpy as np
import torch
import torch.nn as nn

from ..engine import Engine


class Img2SeqFunction(torch.autograd.Function):
    """loss = mean over unmasked tokens of the cross entropy (img2seq.py:68-75); d loss / d params via BPTT."""

    @staticmethod
    def forward(ctx, flat, module, img, formula, lengths):
        eng = module.engine
        module._sync_weights()
        drop = None
        if module.training and 0.0 < module.keep_prob < 1.0:
            module._drop_seed = (module._drop_seed * 1103515245 + 12345) & 0x7FFFFFFF
            drop = (module.keep_prob, module._drop_seed)
        eng.forward(img, formula, dropout=drop)
        n_tok = int(np.asarray(lengths.cpu() if isinstance(lengths, torch.Tensor) else lengths).sum())
        stats = eng.loss(lengths, 1.0 / float(n_tok))                  # device [sum of CE, token count]
        ctx.module = module
        return (stats[0] / stats[1]).reshape(())

    @staticmethod
    def backward(ctx, grad_out):
        eng = ctx.module.engine
        eng.backward()                                                   # BPTT + encoder backward into eng.grads
        return eng.grads * grad_out, None, None, None, None


class Img2Seq(nn.Module):
    """CNN encoder + attention LSTM decoder of the reference (model/encoder.py, model/decoder.py) as one module."""

    def __init__(self, n_tok, dims=None, dtype="bf16", device="cuda:0", seed=0, keep_prob=1.0, lib=None):
        super().__init__()
        self.engine = Engine(n_tok, dims=dims, dtype=dtype, device=device, seed=seed, lib=lib)
        # the parameter shares the engine's buffer: optimizers update the engine in place
        self.flat = nn.Parameter(self.engine.params, requires_grad=True)
        self.engine.params = self.flat.data
        self.keep_prob = float(keep_prob)
        self._drop_seed = int(seed) + 1
        self._packed_version = self.flat._version

    def _sync_weights(self):
        if self.flat._version != self._packed_version:
            self.engine.params = self.flat.data
            self.engine.pack()
            self._packed_version = self.flat._version

    def named_views(self):
        """TF variable name -> view of the flat parameter (SURVEY Appendix B names; checkpoint interchange)."""
        return {k: self.flat.data[o:o + n].view(*s) for k, (o, n, s) in self.engine._offsets.items()}

    def forward(self, img, formula, lengths):
        """img uint8 [B,H,W,1], formula int [B,T] (END-padded), lengths int [B] -> scalar training loss."""
        return Img2SeqFunction.apply(self.flat, self, img, formula, lengths)

    @torch.no_grad()
    def greedy(self, img, id_end, max_iter=151):
        self._sync_weights()
        return self.engine.greedy_decode(img, id_end, max_iter=max_iter)

    @torch.no_grad()
    def beam(self, img, id_end, beam_size, max_iter=151):
        self._sync_weights()
        return self.engine.beam_decode(img, id_end, beam_size, max_iter=max_iter)


class Img2SeqModel(object):
    """Mirror of the hook methods of /root/reference/model/img2seq_torch.py:64-83 and base_torch.py:74-138
    (getModel / getOptimizer / getLoss); the TF-style trainer with the full epoch loop is model/img2seq.py."""

    def __init__(self, config, dir_output, vocab, dtype="bf16", lib=None):
        self._config = config
        self._dir_output = dir_output
        self._vocab = vocab
        self._dtype = dtype
        self._lib = lib
        self.device = torch.device(getattr(config, "device", "cuda:0") if torch.cuda.is_available() else "cpu")
        self.model = None
        self.optimizer = None

    def getModel(self, model_name="Img2Seq"):
        if model_name != "Img2Seq":
            raise NotImplementedError("return your model ({}) that inherits from torch.nn".format(model_name))
        cfg = self._config
        dims = getattr(cfg, "dims", None)
        self.model = Img2Seq(self._vocab.n_tok, dims=dims, dtype=self._dtype, device=str(self.device),
                             keep_prob=float(getattr(cfg, "dropout", 1.0)), lib=self._lib)
        return self.model

    def getOptimizer(self, lr_method="adam", lr=1e-3):
        m = lr_method.lower()
        if m == "adam":
            self.optimizer = torch.optim.Adam(self.model.parameters(), lr=lr)
        elif m == "adamax":
            self.optimizer = torch.optim.Adamax(self.model.parameters(), lr=lr)
        elif m == "sgd":
            self.optimizer = torch.optim.SGD(self.model.parameters(), lr=lr)
        else:
            raise NotImplementedError("Unknown Optimizer {}".format(lr_method))
        return self.optimizer

    def getLoss(self, img, formula, lengths, lr=None, dropout=None, training=True):
        """One batch (reference getLoss, img2seq_torch.py:129-163, without its unfinished attention regulariser):
        forward, and when training: backward + optimizer step.  Returns the loss as a float."""
        if dropout is not None:
            self.model.keep_prob = float(dropout)
        self.model.train(training)
        if lr is not None and self.optimizer is not None:
            for g in self.optimizer.param_groups:
                g["lr"] = lr
        loss = self.model(img, formula, lengths)
        if training:
            self.optimizer.zero_grad()
            loss.backward()
            self.optimizer.step()
        return float(loss.item())
