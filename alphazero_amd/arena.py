"""Arena (arena.py:12-185): games between two Players on one Board, win/loss/draw statistics."""
from collections import defaultdict


class Arena:
    def __init__(self, player1, player2, board):
        self.player1, self.player2, self.board = player1, player2, board
        self.game = board.game

    def play_game(self, player2_starts=False, return_results=False, verbose=False, **_display_kwargs):
        idx = 1 if player2_starts else 0
        self.board.reset()
        self.player1.reset()
        self.player2.reset()
        players = (self.player1, self.player2)
        while not self.board.is_game_over():
            move, _, _, _ = players[idx].get_move(self.board)
            self.board.play_move(move)
            if verbose:
                print(f"{players[idx]} played {move} | score = {-self.board.get_score()}")
            players[idx].apply_move(move, player=-self.board.player)      # every player keeps its own tree
            players[1 - idx].apply_move(move, player=self.board.player)   # (arena.py:98-99)
            idx = 1 - idx
        score = abs(self.board.get_score())
        winner = self.board.get_winner()
        if winner == 0:
            return {"winner": 0, "score": score} if return_results else None
        first_won = (winner == 1 and not player2_starts) or (winner == -1 and player2_starts)
        return {"winner": 1 if first_won else 2, "score": score} if return_results else None

    def play_games(self, n_rounds, start_player=None, return_stats=False, verbose=False, call_id=None):
        if not isinstance(n_rounds, int) or n_rounds < 1:
            raise ValueError("n_rounds must be a positive integer")
        if start_player not in (None, 1, 2):
            raise ValueError("start_player must be None, 1 or 2")
        stats = {"player1": [], "player2": [], "draw": 0,
                 "player1_starts": defaultdict(int), "player2_starts": defaultdict(int)}
        for round_idx in range(n_rounds):
            p2s = {1: False, 2: True}.get(start_player, bool(round_idx % 2))
            res = self.play_game(player2_starts=p2s, return_results=True, verbose=verbose)
            starter = f"player{2 if p2s else 1}_starts"
            if res["winner"] == 0:
                stats["draw"] += 1
                stats[starter]["draw"] += 1
            else:
                stats[f"player{res['winner']}"].append(res["score"])
                stats[starter]["win" if res["winner"] == (2 if p2s else 1) else "loss"] += 1
        return stats if return_stats else None

    def play_games_in_parallel(self, n_rounds, n_process=None, verbose=False, return_stats=False):
        """the reference forks CPU processes here (arena.py:187-257); device trees are not forked -- plays serially"""
        return self.play_games(n_rounds, return_stats=return_stats, verbose=verbose)

    @staticmethod
    def print_stats_results(player1, player2, stats):
        n1, n2, d = len(stats["player1"]), len(stats["player2"]), stats["draw"]
        print(f"{player1} wins: {n1} | {player2} wins: {n2} | draws: {d}")
