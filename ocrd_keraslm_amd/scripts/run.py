# -*- coding: utf-8 -*-
"""`keraslm-rate` command line (drop-in for ocrd_keraslm/scripts/run.py:31-221).

Same commands, options, defaults and outputs as the reference CLI: train, test,
apply, generate, print-history, print-charset, prune-charset and the three plot-* views of the
embeddings (matplotlib / scikit-learn imported on use).
Additional option on `train`: --streams (stateful streams per GPU, default 1 = the
reference's batching).  Under `python -m torch.distributed.run` training is
data-parallel over the launched ranks (lib/distributed.py).
"""
from __future__ import absolute_import

import json
import os
import sys
from math import ceil

import click

from .. import lib

COMMAND_ORDER = ['train', 'test', 'apply', 'generate', 'print-history', 'print-charset', 'prune-charset',
                 'plot-char-embeddings-similarity', 'plot-context-embeddings-similarity', 'plot-context-embeddings-projection']


class OrderedGroup(click.Group):
    def list_commands(self, ctx):
        names = list(super(OrderedGroup, self).list_commands(ctx))
        return [n for n in COMMAND_ORDER if n in names] + sorted(n for n in names if n not in COMMAND_ORDER)


@click.group(cls=OrderedGroup)
def cli():
    pass


def _open_all(items):
    """files, or every regular file of a directory (run.py:70-80, 124-131)"""
    handles = []
    for item in items:
        if os.path.isdir(item):
            paths = [os.path.join(item, name) for name in os.listdir(item)]
            handles.extend(open(p, mode='r') for p in paths if os.path.isfile(p))
        else:
            handles.append(open(item, mode='r'))
    return handles


def _contexts(text):
    """'1784' -> [179]: one context id = ceil(year/10) per blank-separated number (run.py:105-106)"""
    return [ceil(int(x) / 10) for x in text.split(' ')]


def _load(model, incremental=False):
    rater = lib.Rater()
    rater.load_config(model)
    if incremental:
        rater.stateful = False
        rater.incremental = True
    rater.configure()
    rater.load_weights(model)
    return rater


@cli.command(short_help='train a language model')
@click.option('-m', '--model', default="model.h5", show_default=True, help='model file',
              type=click.Path(dir_okay=False, writable=True))
@click.option('-C', '--ckpt', default="ckpt.h5", show_default=True, help='checkpoint file', type=click.Path(dir_okay=False))
@click.option('-w', '--width', default=128, show_default=True, help='number of nodes per hidden layer',
              type=click.IntRange(min=1, max=9128))
@click.option('-d', '--depth', default=2, show_default=True, help='number of hidden layers', type=click.IntRange(min=1, max=10))
@click.option('-l', '--length', default=256, show_default=True, help='number of previous characters seen (window size)',
              type=click.IntRange(min=1, max=1024))
@click.option('-v', '--val-data', default=None, show_default=True,
              help='validation data file or directory (instead of automatic split)',
              type=click.Path(exists=True, dir_okay=True, file_okay=True))
@click.option('-s', '--streams', default=1, show_default=True, help='stateful streams trained in lockstep per GPU',
              type=click.IntRange(min=1, max=4096))
@click.argument('data', nargs=-1, type=click.Path(exists=True, dir_okay=True, file_okay=True))
def train(model, ckpt, width, depth, length, val_data, streams, data):
    """Train a language model from DATA files,
       with parameters WIDTH, DEPTH, and LENGTH.

       The files will be randomly split into training and validation data,
       except if VAL_DATA is given.
    """
    from ..lib.distributed import init_from_env
    rank, _world, _local = init_from_env()
    rater = lib.Rater()
    resume = None
    if os.path.isfile(model):
        rater.load_config(model)
        if rater.width == width and rater.depth == depth:
            resume = model
            print('loading weights from existing model for continued training')
        else:
            print('warning: ignoring existing model due to different topology (width=%d, depth=%d)'
                  % (rater.width, rater.depth), file=sys.stderr)
    elif os.path.isfile(ckpt):
        resume = ckpt
        print('loading weights from checkpoint for continued training')
    rater.width, rater.depth, rater.length = width, depth, length
    rater.streams = streams
    rater.configure()
    if resume:
        if rater.model is None:
            # a bare checkpoint carries no mapping (latent bugs): nothing to resume into
            print('warning: cannot resume from %s without a character mapping' % resume, file=sys.stderr)
        else:
            rater.load_weights(resume)
    training = _open_all(data)
    validation = _open_all([val_data]) if val_data else None
    rater.train(training, val_data=validation)
    assert rater.status == 2
    if rank == 0:
        rater.save(model)


@cli.command(short_help='get individual probabilities from language model')
@click.option('-m', '--model', required=True, help='model file', type=click.Path(dir_okay=False, exists=True))
@click.option('-c', '--context', default=None, help='constant meta-data input')
@click.argument('text', type=click.STRING)
def apply(model, text, context):
    """Apply a language model to TEXT string and compute its individual probabilities.

       If TEXT is the symbol '-', the string will be read from standard input.
    """
    rater = _load(model)
    if text and text[0] == u"-":
        text = sys.stdin.read()
    ratings, perplexity = rater.rate2(text, _contexts(context) if context else None)
    click.echo(perplexity)
    click.echo(json.dumps(ratings, ensure_ascii=False))


@cli.command(short_help='get overall perplexity from language model')
@click.option('-m', '--model', required=True, help='model file', type=click.Path(dir_okay=False, exists=True))
@click.argument('data', nargs=-1, type=click.Path(exists=True, dir_okay=True, file_okay=True))
def test(model, data):
    """Apply a language model to DATA files and compute its overall perplexity."""
    rater = _load(model)
    click.echo(rater.test(_open_all(data)))


@cli.command(short_help='sample characters from language model')
@click.option('-m', '--model', required=True, help='model file', type=click.Path(dir_okay=False, exists=True))
@click.option('-n', '--number', default=1, help='number of characters to sample', type=click.IntRange(min=1, max=10000))
@click.option('-v', '--variants', default=1, help='number of character sequences to sample',
              type=click.IntRange(min=1, max=10000))
@click.option('-c', '--context', default=None, help='constant meta-data input')
@click.argument('prefix', type=click.STRING)
def generate(model, number, variants, context, prefix):
    """Apply a language model, generating the most probable characters (starting with PREFIX string)."""
    rater = _load(model, incremental=True)
    ctx = _contexts(context) if context else rater.underspecify_contexts()
    for res in rater.generate(prefix, number, ctx, variants):
        click.echo(prefix[:-1] + res)


@cli.command(short_help='Print the training history')
@click.option('-m', '--model', required=True, help='model file', type=click.Path(dir_okay=False, exists=True))
def print_history(model):
    rater = lib.Rater()
    rater.load_config(model)
    rater.print_history()


@cli.command(short_help='Print the mapped characters')
@click.option('-m', '--model', required=True, help='model file', type=click.Path(dir_okay=False, exists=True))
def print_charset(model):
    rater = lib.Rater()
    rater.load_config(model)
    rater.print_charset()


@cli.command(short_help='Delete one character from mapping')
@click.option('-m', '--model', required=True, help='model file', type=click.Path(dir_okay=False, exists=True, writable=True))
@click.argument('char')
def prune_charset(model, char):
    rater = _load(model)
    if rater.remove_from_mapping(char=char):
        rater.save(model)


@cli.command(short_help='Paint a heat map of character embeddings')
@click.option('-m', '--model', required=True, help='model file', type=click.Path(dir_okay=False, exists=True))
@click.argument('filename', type=click.Path(dir_okay=False, writable=True))
def plot_char_embeddings_similarity(model, filename):
    _load(model).plot_char_embeddings_similarity(filename)


@cli.command(short_help='Paint a heat map of context embeddings')
@click.option('-m', '--model', required=True, help='model file', type=click.Path(dir_okay=False, exists=True))
@click.option('-n', '--number', default=1, help='which context variable', type=click.IntRange(min=1, max=100))
@click.argument('filename', type=click.Path(dir_okay=False, writable=True))
def plot_context_embeddings_similarity(model, filename, number):
    _load(model).plot_context_embeddings_similarity(filename, n=number)


@cli.command(short_help='Paint a 2-d PCA projection of context embeddings')
@click.option('-m', '--model', required=True, help='model file', type=click.Path(dir_okay=False, exists=True))
@click.option('-n', '--number', default=1, help='which context variable', type=click.IntRange(min=1, max=100))
@click.argument('filename', type=click.Path(dir_okay=False, writable=True))
def plot_context_embeddings_projection(model, filename, number):
    _load(model).plot_context_embeddings_projection(filename, n=number)


if __name__ == '__main__':
    cli()
