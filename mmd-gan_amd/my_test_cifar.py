"""python my_test_cifar.py [--synthetic] [--steps N] [--rounds N] [--loss rep|rmb|...] - see drivers.py"""
from drivers import run

if __name__ == '__main__':
    run('cifar')
