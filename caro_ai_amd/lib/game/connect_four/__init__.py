"""`lib.game.connect_four` of the reference is a package; so is this one, and
`from caro_ai_amd.lib.game.connect_four import ConnectFour` keeps working."""
from caro_ai_amd.lib.game.connect_four.connect_four import ConnectFour  # noqa: F401
