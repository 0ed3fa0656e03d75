"""``parse_generate_args``: the flag names and defaults of the reference's
``utilities/argument_generate_funcs.py:34-104`` that reach the AMT hot path, plus ``--synthetic``
inputs (the reference tree ships neither weights nor dataset features)."""
import argparse

from .constants import IS_VIDEO, RPR, VERSION


def parse_generate_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("-dataset_dir", type=str, default="./dataset/", help="Folder of VEVO dataset")
    parser.add_argument("-output_dir", type=str, default="./output_vevo/" + VERSION, help="Folder to write generated chords to")
    parser.add_argument("-primer_file", type=str, default=None)
    parser.add_argument("--force_cpu", action="store_true", help="Kept for flag compatibility; this build has no CPU path")
    parser.add_argument("-target_seq_length_chord", type=int, default=300, help="Target length of the chord sequence")
    parser.add_argument("-num_prime_chord", type=int, default=30)
    parser.add_argument("-model_weights", type=str, default="saved_models/AMT/best_loss_weights.pickle",
                        help="state_dict saved with torch.save (reference key names)")
    parser.add_argument("-beam", type=int, default=0, help="0 for random probability sample and 1 for greedy")
    parser.add_argument("-max_sequence_midi", type=int, default=2048)
    parser.add_argument("-max_sequence_video", type=int, default=300)
    parser.add_argument("-max_sequence_chord", type=int, default=300)
    parser.add_argument("-chord_embed", type=bool, default=True)
    parser.add_argument("-n_layers", type=int, default=6)
    parser.add_argument("-num_heads", type=int, default=8)
    parser.add_argument("-d_model", type=int, default=512)
    parser.add_argument("-dim_feedforward", type=int, default=1024)
    parser.add_argument("-rms_norm", type=bool, default=False)
    parser.add_argument("-music_gen_version", type=str, default="2.2",
                        help="'2.2' (reference default): VideoMusicTransformer_V2; 'None': the base AMT (VideoMusicTransformer, "
                             "the batched KV-cached fast path)")
    parser.add_argument("-scene_embed", type=bool, default=False)
    parser.add_argument("-is_video", type=bool, default=IS_VIDEO)
    parser.add_argument("-emo_model", type=str, default="6c_l14p")
    parser.add_argument("-motion_type", type=int, default=1, help="0 as original, 1 as option 1, 2 as option 2")
    parser.add_argument("-rpr", type=bool, default=RPR)
    # regression head (argument_generate_funcs.py:64,87-91)
    parser.add_argument("-modelReg_weights", type=str, default="saved_models/AMT/best_rmse_weights.pickle")
    parser.add_argument("-n_layers_reg", type=int, default=6)
    parser.add_argument("-d_model_reg", type=int, default=128)
    parser.add_argument("-dim_feedforward_reg", type=int, default=256)
    parser.add_argument("-regModel", type=str, default="bimamba+")
    # additions of this build
    parser.add_argument("--synthetic", action="store_true", help="random-init procedural weights and random video features")
    parser.add_argument("--n_clips", type=int, default=1, help="clips to generate for (sharded over ranks under torchrun)")
    parser.add_argument("--test_ids", type=str, default=None,
                        help="clip ids to read from -dataset_dir, comma separated, or split:<name> for vevo_meta/split/v1/<name>.txt "
                             "(the reference hard-codes one test_id at generate.py:29)")
    parser.add_argument("--primer", type=str, default=None,
                        help='custom primer chords as a user types them, e.g. "C Am Dm G" (the reference edits isPrimer / custumPrimer '
                             'in generate.py:46-53 for this); default: "C" or "A:min" by the clip\'s key')
    parser.add_argument("--primer_from_dataset", action="store_true",
                        help="prime with the clip's own first -num_prime_chord chords (generate.py:367-379: isPrimer with an empty custumPrimer)")
    parser.add_argument("--synthetic_weights", action="store_true", help="random-init procedural weights with real feature files")
    parser.add_argument("--regression", action="store_true",
                        help="also run the VideoRegression head and write <id>_loudness_density.csv (generate.py:394-409)")
    parser.add_argument("--midi", action="store_true",
                        help="also write <id>_chords.mid (voiced arpeggios, velocities from the regression head when --regression is set; "
                             "generate.py:446-607)")
    parser.add_argument("--v2_batch", type=int, default=32, help="clips of a V1 / V2 run decoded together in lockstep (one captured step graph)")
    parser.add_argument("--sampler", type=str, default="categorical", choices=["categorical", "multinomial", "argmax"],
                        help="beam=0 decision: categorical = on-device draw (default), multinomial = host torch.multinomial per step, argmax")
    parser.add_argument("--seed", type=int, default=1234)
    return parser.parse_known_args(argv)
