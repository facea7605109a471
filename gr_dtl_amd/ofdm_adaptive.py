"""Frame-size helpers of the adaptive OFDM link (interface of python/dtl/ofdm_adaptive.py:1-8: `frame_capacity`,
`max_bps`).  A frame of L OFDM payload symbols walks cyclically through the configured sets of data carriers;
its capacity is the number of constellation symbols that fit."""


def frame_capacity(n_ofdm_syms, occupied_carriers):
    """Constellation symbols carried by `n_ofdm_syms` OFDM symbols when symbol i uses carrier set i mod #sets."""
    sets = [len(carriers) for carriers in occupied_carriers]
    if not sets or n_ofdm_syms <= 0:
        return 0
    whole, rest = divmod(int(n_ofdm_syms), len(sets))
    return whole * sum(sets) + sum(sets[:rest])


def max_bps(cnsts):
    """Bits per symbol of the densest constellation in `cnsts` (constellation ids equal their bits per symbol)."""
    return max(int(c) for c in cnsts)
