"""Speaker naming / track-order tables (same values as reference core/constants.py:5-104)."""

SPEAKER_NAMES = ['FL', 'FR', 'FC', 'BL', 'BR', 'SL', 'SR', 'WL', 'WR', 'TFL', 'TFR', 'TSL', 'TSR', 'TBL', 'TBR']

LEFT_SIDE_SPEAKERS = frozenset(n for n in SPEAKER_NAMES if n.endswith('L'))
RIGHT_SIDE_SPEAKERS = frozenset(n for n in SPEAKER_NAMES if n.endswith('R'))

TRUEHD_11CH_ORDER = ['FL', 'FR', 'FC', 'BL', 'BR', 'SL', 'SR', 'TFL', 'TFR', 'TBL', 'TBR']
TRUEHD_13CH_ORDER = ['FL', 'FR', 'FC', 'BL', 'BR', 'SL', 'SR', 'TFL', 'TFR', 'TSL', 'TSR', 'TBL', 'TBR']

# delays relative to the nearest speaker: all zero in the reference (core/constants.py:59-61)
SPEAKER_DELAYS = {name: 0 for name in SPEAKER_NAMES}

IPSILATERAL_PAIRS = (("FL", "FR"), ("SL", "SR"), ("BL", "BR"), ("TFL", "TFR"), ("TSL", "TSR"),
                     ("TBL", "TBR"), ("FC", "FC"), ("WL", "WR"))

SEQUENCE_TRACK_ORDERS = {
    '5.1': 'FL FR FC LFE BL BR'.split(),
    '7.1': 'FL FR FC LFE BL BR SL SR'.split(),
    '7.1.4': 'FL FR FC LFE BL BR SL SR TFL TFR TBL TBR'.split(),
    '7.1.6': 'FL FR FC LFE BL BR SL SR TFL TFR TSL TSR TBL TBR'.split(),
}


def _order(spec):
    return [f"{sp}-{sd}" for sp, sd in (item.split(':') for item in spec.split())]


HESUVI_TRACK_ORDER = _order(
    "FL:left FL:right SL:left SL:right BL:left BL:right FC:left FR:right FR:left SR:right SR:left BR:right "
    "BR:left FC:right WL:left WL:right WR:left WR:right TFL:left TFL:right TFR:left TFR:right TSL:left "
    "TSL:right TSR:left TSR:right TBL:left TBL:right TBR:left TBR:right")

HEXADECAGONAL_TRACK_ORDER = [f"{sp}-{sd}" for sp in
                             "FL FR FC LFE BL BR SL SR WL WR TFL TFR TSL TSR TBL TBR".split()
                             for sd in ("left", "right")]


def speaker_side(name):
    """'left' / 'right' / 'center' for a speaker name (reference core/constants.py:19-26)."""
    up = name.upper()
    if up in LEFT_SIDE_SPEAKERS:
        return 'left'
    if up in RIGHT_SIDE_SPEAKERS:
        return 'right'
    return 'center'


def track_name(speaker, side):
    return f"{speaker}-{side}"
