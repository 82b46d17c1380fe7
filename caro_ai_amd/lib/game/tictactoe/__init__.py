"""`lib.game.tictactoe` of the reference is a package (tictactoe.py + tictactoe_helpers.py); so is this one, and
`from caro_ai_amd.lib.game.tictactoe import TicTacToe` keeps working."""
from caro_ai_amd.lib.game.tictactoe.tictactoe import TicTacToe  # noqa: F401
