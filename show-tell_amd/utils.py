"""Checkpoint wire format and caption helpers of the reference's ``utils.py``.

``create_checkpoint`` writes exactly the dictionary the reference writes (utils.py:125-145):
``{'encoder_state_dict', 'decoder_state_dict', 'optimizer_state_dict', 'epoch', 'step'}`` to
``<output_dir>/model_<epoch>.ckpt`` plus ``model_<epoch>_metrics.ckpt = {'train_loss': [...]}``; because
the modules keep the reference's ``state_dict`` keys and the optimizers write torch.optim's per-parameter layout
(optim.py), checkpoints move both ways between the two code bases.
``load_checkpoint`` mirrors main.py:117-123 / utils.py:151-154.  No kernels are involved: this is host I/O.
"""
import os

import torch


def create_checkpoint(cnn, rnn, optimizer, epoch, step, train_loss, params, trainer=None):
    '''Function to create a checkpoint for the trained models and their corresponding evaluated metrics (utils.py:125).
    `trainer`: a showtell_amd.train.Trainer whose last optimizer.step() may still be pending (it is applied after the NEXT
    step's frozen backbone so that the gradient all-reduce hides behind it); it is flushed first so that the file holds what
    the reference's loop would hold at this point.'''
    if trainer is None:
        trainer = getattr(optimizer, '_pending_owner', None)   # a Trainer registers itself on its optimizer (train.py): the
    if trainer is not None:                                    # reference call signature flushes the deferred update too
        trainer.flush()
    model_file = 'model_' + str(epoch) + '.ckpt'
    metrics_file = 'model_' + str(epoch) + '_metrics.ckpt'
    os.makedirs(params['output_dir'], exist_ok=True)
    torch.save({'encoder_state_dict': {k: v.detach().cpu() for k, v in cnn.state_dict().items()},
                'decoder_state_dict': {k: v.detach().cpu() for k, v in rnn.state_dict().items()},
                'optimizer_state_dict': optimizer.state_dict(),
                'epoch': epoch,
                'step': step},
               os.path.join(params['output_dir'], model_file))
    torch.save({'train_loss': train_loss}, os.path.join(params['output_dir'], metrics_file))
    return os.path.join(params['output_dir'], model_file)


def load_checkpoint(path, cnn, rnn, optimizer=None, map_location='cpu'):
    """main.py:117-123: restores encoder / decoder (and optimizer) state; returns (epoch, step)."""
    sd = torch.load(path, map_location=map_location, weights_only=True)
    cnn.load_state_dict(sd['encoder_state_dict'])
    rnn.load_state_dict(sd['decoder_state_dict'])
    if optimizer is not None and sd.get('optimizer_state_dict'):
        optimizer.load_state_dict(sd['optimizer_state_dict'])   # main.py:121; a mismatching checkpoint raises, as torch.optim's does
    return sd.get('epoch', 0), sd.get('step', 0)


def create_caption_word_format(tokenized, vocab, flag_blue=False):
    '''ids -> words, stopping at <end> and skipping <start> (utils.py:105-123).'''
    caption_words = []
    for token in tokenized:
        curr_word = []
        for idx in token:
            idx = int(idx)
            if vocab.index_to_word[idx] == vocab.end_token():
                break
            if idx != vocab.word_to_index[vocab.start_token()]:
                curr_word.append(vocab.index_to_word[idx])
        caption_words.append([curr_word] if flag_blue else curr_word)
    return caption_words
