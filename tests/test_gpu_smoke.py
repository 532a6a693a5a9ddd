import pytest

pytestmark = pytest.mark.gpu


def test_graft_smoke():
    import __graft_entry__ as g
    g.build()
    g.smoke()
