from numpy.testing import (  # noqa: F401
    assert_equal, assert_almost_equal, assert_array_almost_equal, assert_allclose)
