from ...distributions import GumbelSoftmaxBernoulliStraightThrough  # noqa: F401
