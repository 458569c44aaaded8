bash tools/final_round.sh r04
