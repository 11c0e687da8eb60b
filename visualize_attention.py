#!/usr/bin/env python
"""Attention visualisation with the shape of the reference's visualize_attention.py:22-178: one PNG per decoded
token (the attention map blended over the input image) and an animated GIF.  The reference collects the maps in a
Python global through a tf.py_func inside the graph; here they come back from the decode call
(lxo_greedy_decode_attn / Engine.greedy_decode(return_attention=True)).  Rendering uses PIL only (the reference's
matplotlib/imagemagick animation writer is replaced by PIL's GIF writer)."""
import argparse
import os

import numpy as np
from PIL import Image, ImageDraw

from latex_ocr_amd.model.utils.image import encoder_out_hw, greyscale


def getWH(img_w, img_h):
    """visualize_attention.py:22-31."""
    h, w = encoder_out_hw(img_h, img_w)
    return w, h


def getOutArray(attentionVector, att_w, att_h):
    """visualize_attention.py:48-73: [att_h, att_w] grey map, darker = more attention ((1 - a) * 255)."""
    a = np.asarray(attentionVector, np.float64).reshape(-1)[:att_w * att_h]
    return ((1.0 - a) * 255.0).reshape(att_h, att_w)


def getCombineArray(attentionVector, inp_image, img_w, img_h, att_w, att_h):
    """visualize_attention.py:82-88: nearest-neighbour upsample of the map, 50 % blend over the image."""
    out_image = Image.fromarray(getOutArray(attentionVector, att_w, att_h)).resize((img_w, img_h), Image.NEAREST)
    return np.asarray(Image.blend(inp_image.convert("RGBA"), out_image.convert("RGBA"), 0.5))


def getFileNameToSave(path_to_save_attention, i):
    return path_to_save_attention + "_" + str(i) + ".png"


def vis_attention_slices(inp_image, alphas, path_to_save_attention):
    """visualize_attention.py:33-45: one PNG per decoding step."""
    img_w, img_h = inp_image.size
    att_w, att_h = getWH(img_w, img_h)
    files = []
    for i, a in enumerate(alphas):
        fn = getFileNameToSave(path_to_save_attention, i)
        Image.fromarray(getCombineArray(a, inp_image, img_w, img_h, att_w, att_h)).save(fn)
        files.append(fn)
    return files


def vis_attention_gif(inp_image, alphas, path_to_save_attention, hyp, full_latex=False):
    """visualize_attention.py:117-160: frame i = blend of step i with the tokens decoded so far underneath
    (the current one in red)."""
    img_w, img_h = inp_image.size
    att_w, att_h = getWH(img_w, img_h)
    symbols = hyp.split(" ")
    frames = []
    for i in range(min(len(symbols), len(alphas))):
        comb = Image.fromarray(getCombineArray(alphas[i], inp_image, img_w, img_h, att_w, att_h))
        frame = Image.new("RGBA", (img_w, img_h + 16), (255, 255, 255, 255))
        frame.paste(comb, (0, 0))
        d = ImageDraw.Draw(frame)
        x = 2
        for j, s in enumerate(symbols if full_latex else symbols[:i + 1]):
            d.text((x, img_h + 2), s, fill=(255, 0, 0, 255) if j == i else (0, 128, 0, 255))
            x += 6 * (len(s) + 1)
        frames.append(frame.convert("P"))
    fn = path_to_save_attention + "_visualization1.gif"
    if frames:
        frames[0].save(fn, save_all=True, append_images=frames[1:], duration=200, loop=0)
    return fn


def vis_img_with_attention(img2SeqModel, img_path, dir_output):
    """visualize_attention.py:181-196."""
    inp_image = Image.open(img_path)
    img = greyscale(np.asarray(inp_image.convert("RGB")))
    hyp, alphas = img2SeqModel.predict_with_attention(img)
    print(hyp)
    os.makedirs(dir_output + "vis/", exist_ok=True)
    path = dir_output + "vis/vis_" + os.path.basename(img_path)[:-4]
    files = vis_attention_slices(inp_image, alphas, path)
    files.append(vis_attention_gif(inp_image, alphas, path, hyp))
    return hyp, files


def main(argv=None):
    from latex_ocr_amd.model.img2seq import Img2SeqModel
    from latex_ocr_amd.model.utils.general import Config
    from latex_ocr_amd.model.utils.text import Vocab
    ap = argparse.ArgumentParser()
    ap.add_argument("--image", default="data/images_test/6.png")
    ap.add_argument("--results", default="results/small/")
    a = ap.parse_args(argv)
    d = a.results
    model = Img2SeqModel(Config(d + "model.json"), d, Vocab(Config(d + "vocab.json")))
    model.build_pred()
    return vis_img_with_attention(model, a.image, d)


if __name__ == "__main__":
    main()
