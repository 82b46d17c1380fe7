"""MI355X-native AlphaZero self-play engine behind the caro-ai plugin API.

Only the self-play hot path lives here: batched game
rules, the MCTS select / expand / backup walk and the game loop as HIP kernels
(caro_ai_amd/csrc, C-ABI in include/caro_hip.h), plus the host-side mirror of
the reference's `BaseGame` / `MCTS` / `play_game` / `Net` interface.
"""
__version__ = "0.1.0"
