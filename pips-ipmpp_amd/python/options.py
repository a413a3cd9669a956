"""The reference's options file (PIPSIPMpp.opt: lines `IDENTIFIER value type`, type in bool/boolean, int/integer, double;
`#` and `//` comments; AbstractOptions::load_options_from_file, Core/Options/AbstractOptions.C:62-135) and the mapping of the
identifiers that concern the KKT path onto this library's settings.  Identifiers that steer other subsystems of the reference
(presolve, scaling, hierarchical approach, ...) are parsed and reported as ignored."""

_BOOL = {"true": True, "TRUE": True, "True": True, "false": False, "FALSE": False, "False": False}


def load_options(path):
    """{identifier: value}; malformed lines are skipped like the reference does (it prints a warning and continues)."""
    opts = {}
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line or (len(line) > 1 and line[0] == "#") or (len(line) > 2 and line[:2] == "//"):
                continue
            parts = line.split()
            if len(parts) < 3 or parts[0][0] == "#":
                continue
            ident, value, typ = parts[0], parts[1], parts[2]
            try:
                if typ in ("int", "integer"):
                    opts[ident] = int(value)
                elif typ == "double":
                    opts[ident] = float(value)
                elif typ in ("bool", "boolean") and value in _BOOL:
                    opts[ident] = _BOOL[value]
            except ValueError:
                continue
    return opts


def apply_options(opts, batch=None, ipm=None):
    """Applies the path-relevant identifiers; returns (applied, ignored) lists of identifiers.

    batch (LeafBatch, before analyze):
      SC_COMPUTE_BLOCKWISE        true -> Schur mode 2 (blocked multi-RHS solves, the reference's K4-K6 loop); false -> mode 1
                                  (partial factorisation of the augmented block - what PardisoSchurSolver does in the reference)
      PARDISO_NITERATIVE_REFINS   >= 0 -> at most that many refinement steps per leaf solve (iparm[7])
      PARDISO_PIVOT_PERTURBATION  k > 0 -> pivots replaced at 1e-k relative (iparm[9])
    ipm (IpmSolver / GeneralIpmSolver): GONDZIO_MAX_CORRECTORS, OUTER_SOLVE, OUTER_BICG_MAX_ITER, OUTER_BICG_MAX_NORMR_DIVERGENCES,
      OUTER_BICG_MAX_STAGNATIONS, OUTER_SOLVE_REFINE_ORIGINAL_SYSTEM (false = every outer solve on the regularised system, what
      PIPSIPMppOptions.C:293 sets), REGULARIZATION"""
    applied, ignored = [], []
    for ident, value in opts.items():
        done = False
        if batch is not None:
            if ident == "SC_COMPUTE_BLOCKWISE":
                batch.set_schur_mode(2 if value else 1)
                done = True
            elif ident == "PARDISO_NITERATIVE_REFINS" and value >= 0:
                batch.set_refinement(int(value), 0.0)
                done = True
            elif ident == "PARDISO_PIVOT_PERTURBATION" and value > 0:
                batch.set_options(repl_rel=10.0 ** (-int(value)))
                done = True
        if ipm is not None and ident in ("GONDZIO_MAX_CORRECTORS", "OUTER_SOLVE", "OUTER_BICG_MAX_ITER", "OUTER_BICG_MAX_NORMR_DIVERGENCES",
                                         "OUTER_BICG_MAX_STAGNATIONS", "OUTER_BICG_EPSILON", "OUTER_SOLVE_REFINE_ORIGINAL_SYSTEM", "REGULARIZATION"):
            if ident == "OUTER_SOLVE" and value == 0:
                ignored.append(ident)   # the harness always refines against the original system
                continue
            ipm.set_option(ident, float(value))
            done = True
        (applied if done else ignored).append(ident)
    return applied, ignored
