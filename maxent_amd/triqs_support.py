"""What the reference's ``triqs_support`` module answers when TRIQS is not installed (reference
python/triqs_support.py.in:23-95): this package never depends on TRIQS -- G(tau) comes in as arrays or text files --
so scripts that ask get "no TRIQS", and functions that need it say so."""

import functools


def if_triqs_2():
    return False


def if_triqs_1():
    return False


def if_no_triqs():
    return True


def require_triqs(func):
    """decorator: the function needs TRIQS Green-function objects"""
    @functools.wraps(func)
    def needs_triqs(*args, **kwargs):
        raise NotImplementedError('{} needs TRIQS, which this package does not use; hand over G(tau) as arrays '
                                  '(set_G_tau_data) or text files (set_G_tau_file)'.format(func.__name__))
    return needs_triqs


def assert_text_files_equal(fname1, fname2):
    with open(fname1, 'rb') as a, open(fname2, 'rb') as b:
        assert a.read() == b.read(), 'files {} and {} are not the same'.format(fname1, fname2)
