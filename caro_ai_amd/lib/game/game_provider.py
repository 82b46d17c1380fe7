"""`-g 0/1` game selection, as lib/game/game_provider.py:5-22 of the reference."""
from caro_ai_amd.lib.game.connect_four import ConnectFour
from caro_ai_amd.lib.game.tictactoe import TicTacToe


def add_game_argument(parser):
    parser.add_argument("-g", "--game", required=True, choices=["0", "1"],
                        help="The type of game. 0: Connect4, 1: TicTacToe")


def get_game(args):
    return ConnectFour() if args.game == "0" else TicTacToe()
