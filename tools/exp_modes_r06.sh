# round 6: what decides the headline's per-process mode? (profiles/r06_headline_modes.txt)
echo "## A: four arenas alive together, per process (5 fresh processes)"
for i in 1 2 3 4 5; do python tools/exp_arena_candidates.py 4 2>/dev/null; echo; done
echo "## B: address-space randomisation off (setarch -R), 6 fresh processes, one arena"
for i in 1 2 3 4 5 6; do setarch x86_64 -R python tools/exp_arena_candidates.py 1 2>/dev/null | head -1; done
echo "## C: randomisation on, 6 fresh processes, one arena"
for i in 1 2 3 4 5 6; do python tools/exp_arena_candidates.py 1 2>/dev/null | head -1; done
